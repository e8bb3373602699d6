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
FILES = sorted(glob.glob(os.path.join(CAP, "aviary_*.npz")))
DOGFIGHT_FILES = sorted(glob.glob(os.path.join(CAP, "env_dogfight_*.npz")))
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


@pytest.mark.skipif(not DOGFIGHT_FILES, reason="no real-PyBullet dogfight capture under tests/golden/pybullet/")
@pytest.mark.parametrize("path", DOGFIGHT_FILES or [None])
def test_oracle_dogfight_against_real_pybullet(path):
    """MAFixedwingDogfightEnv recorded from the real stack (motor noise zeroed) against orc_dogfight_*: everything up to the first
    contact anywhere in the world (after it the contact model is ours, not Bullet's)."""
    g = np.load(path)
    W = O.OracleDogfight(g["start_pos"], g["start_orn"], noise_mode=O.NOISE_OFF, team_size=int(g["team_size"]),
                         damage_per_hit=float(g["damage_per_hit"]), lethal_distance=float(g["lethal_distance"]), lethal_angle=float(g["lethal_angle"]),
                         aggressiveness=float(g["aggressiveness"]), cooperativeness=float(g["cooperativeness"]), sparse_reward=bool(g["sparse_reward"]),
                         dome=float(g["dome"]), max_duration_seconds=int(g["max_steps"]) / 30.0, assisted_flight=int(g["action_dim"]) == 4)
    np.testing.assert_allclose(W.reset(), g["reset_obs"], rtol=1e-6, atol=1e-6)
    first_contact = int(np.argmax(g["contact"])) if g["contact"].any() else len(g["action"])
    for k in range(min(first_contact, len(g["action"]))):
        alive = g["alive"][k]
        obs, rew, term, trunc = W.step(g["action"][k])
        for i in range(W.A):
            if alive[i]:
                np.testing.assert_allclose(obs[i], g["obs"][k][i], rtol=1e-6, atol=1e-6, err_msg=f"step {k} agent {i}")
                assert abs(rew[i] - g["reward"][k][i]) <= 1e-5 * max(1.0, abs(g["reward"][k][i]))
                assert bool(term[i]) == bool(g["term"][k][i]) and bool(trunc[i]) == bool(g["trunc"][k][i])
        np.testing.assert_allclose(W.health, g["health"][k], atol=1e-6)
