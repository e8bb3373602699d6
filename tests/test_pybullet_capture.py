"""Consumes tests/golden/pybullet/*.npz -- trajectories recorded from the REAL PyFlyt + pybullet stack by
tests/golden/capture_pybullet.py on a machine that has them -- when such files are present, and skips otherwise
(neither package exists in the build container or on the GPU boxes: the Bullet boundary is "parity unpinned"
until someone runs the capture script).

  * CPU: the fp64 oracle against real Bullet, 1e-6 (two fp64 implementations of the same recurrences).
  * GPU (-m gpu): the HIP path against real Bullet, 1e-4 -- north_star's bar, literally.
Both compare up to (not including) the first reported contact: what happens after an impact depends on Bullet's
contact solver, which the restated contact response (a named-parameter model) does not claim to reproduce digit
for digit. The step at which the contact is first REPORTED must agree (it pins `contact_report_distance`)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

CAP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pybullet")
FILES = sorted(glob.glob(os.path.join(CAP, "*.npz")))
needs_capture = pytest.mark.skipif(not FILES, reason="no real-PyBullet capture under tests/golden/pybullet/ "
                                                      "(run tests/golden/capture_pybullet.py where PyFlyt + pybullet are installed)")


def model_of(name):
    if "acrowing" in name:
        return "acrowing"
    if "rocket" in name:
        return "rocket"
    return "fixedwing" if "fixedwing" in name else ("primitive_drone" if "primitive" in name else "quadx")


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_capture_script_is_committed():
    assert os.path.exists(os.path.join(os.path.dirname(CAP), "capture_pybullet.py"))


@needs_capture
@pytest.mark.parametrize("path", FILES or [None])
def test_oracle_against_real_pybullet(path):
    g = np.load(path)
    name = os.path.basename(path)[:-4]
    model = model_of(name)
    extra = dict(starting_fuel_ratio=float(g["starting_fuel_ratio"])) if model == "rocket" else {}
    P = O.make_params(model, noise_mode=O.NOISE_OFF, start_pos=g["start_pos"], start_rpy=g["start_orn"], **extra)
    L = O.Lane()
    lib = O.lib()
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    lib.orc_set_mode(C.byref(P), C.byref(L), int(g["mode"]))
    first_contact = int(np.argmax(g["contact"])) if g["contact"].any() else len(g["states"])
    worst = 0.0
    for k in range(len(g["states"])):
        for i, x in enumerate(g["setpoints"][k]):
            L.setpoint[i] = x
        lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
        if k <= first_contact:
            assert bool(L.contact_step) == bool(g["contact"][k]), (name, k, "contact report step")
        if k >= first_contact:
            break
        st = np.array([list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)])
        scale = np.maximum(1.0, np.linalg.norm(g["states"][k], axis=1, keepdims=True))
        worst = max(worst, float((np.abs(st - g["states"][k]) / scale).max()))
    print(f"{name}: oracle vs real pybullet, worst {worst:.2e} over {min(first_contact, len(g['states']))} steps")
    assert worst < 1e-6, worst


@needs_capture
@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES or [None])
def test_hip_against_real_pybullet(path):
    import torch

    import test_gpu_golden as tg

    g = np.load(path)
    name = os.path.basename(path)[:-4]
    if "rocket" in name:
        tg.ROCKET_FUEL[name] = float(g["starting_fuel_ratio"])
    vehicle, eng = tg.aviary_engine(name, g)
    spn = g["setpoints"].shape[1]
    sp = torch.zeros(tg.N, spn, dtype=torch.float32, device=tg.DEV)
    eng.aviary_set_mode(int(g["mode"]), sp)
    first_contact = int(np.argmax(g["contact"])) if g["contact"].any() else len(g["states"])
    worst = 0.0
    for k in range(len(g["states"])):
        sp.copy_(torch.tensor(np.repeat(g["setpoints"][k][None], tg.N, axis=0), dtype=torch.float32))
        eng.aviary_step(sp, 1)
        if k <= first_contact:
            assert (eng.out_contact.cpu().numpy() == bool(g["contact"][k])).all(), (name, k)
        if k >= first_contact:
            break
        worst = max(worst, tg.state_err(eng, g["states"][k], g["aux"][k]))
    print(f"{name}: HIP vs real pybullet, worst {worst:.2e}")
    assert worst < 1e-4, worst
