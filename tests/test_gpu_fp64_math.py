"""The cascaded controller's double-precision helpers on the device (pyflyt_amd/csrc/quadx_control_d.hpp, round 6: a float32 estimate
corrected in fp64 instead of the math library's atan2 / asin / cos / sin, Newton-refined v_rcp_f64 / v_rsq_f64 instead of IEEE division
and square root) against numpy's float64 functions, on the ranges the controller feeds them and well beyond. The bound is 'a few units
in 1e-16': what DESIGN.md section 3 states, five orders below what the cascade amplifies into 1e-4."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    so = tmp_path_factory.mktemp("probe") / "libfp64probe.so"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "device_probes", "fp64_math_probe.hip"), "-o", str(so)])
    lib = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    lib.probe.argtypes = [dp] * 8 + [C.c_int]
    lib.probe.restype = C.c_int

    def run(y, x):
        y, x = np.ascontiguousarray(y, np.float64), np.ascontiguousarray(x, np.float64)
        outs = [np.empty_like(x) for _ in range(6)]
        rc = lib.probe(*(a.ctypes.data_as(dp) for a in (y, x, *outs)), len(x))
        assert rc == 0, rc
        return outs

    return run


@pytest.mark.gpu
def test_atan2_d_rcp_d_sqrt_pos_d(probe):
    rng = np.random.default_rng(5)
    n = 1 << 18
    # the controller's arguments are products of unit-quaternion components: magnitudes in [0.0045, 2]; the test goes to 1e-6 .. 1e3 and
    # adds the axes, the diagonals and both sides of the branch cut
    mag = 10.0 ** rng.uniform(-6, 3, n)
    ang = rng.uniform(-np.pi, np.pi, n)
    y, x = mag * np.sin(ang), mag * np.cos(ang)
    special = np.array([[0.0, 1.0], [0.0, -1.0], [1.0, 0.0], [-1.0, 0.0], [1.0, 1.0], [-1.0, 1.0], [1.0, -1.0], [-1.0, -1.0], [1e-9, -1.0], [-1e-9, -1.0],
                        [-0.0, -1.0], [0.0, 0.0], [1e-12, 1.0], [1.0, 1e-12], [3.0, 4.0]])
    y, x = np.concatenate([y, special[:, 0]]), np.concatenate([x, special[:, 1]])
    at, cs, sn, rc, sq, asn = probe(y, x)
    ref = np.arctan2(y, x)
    e_at = np.abs(at - ref)
    assert e_at.max() < 1e-15, (e_at.max(), y[e_at.argmax()], x[e_at.argmax()])  # (pi itself is known to 4e-16 in a double)
    assert np.signbit(at[len(mag) + 10]) and abs(at[len(mag) + 10] + np.pi) < 1e-15  # atan2(-0, -1) = -pi: the branch cut's lower side
    assert at[len(mag) + 11] == 0.0  # atan2(0, 0)
    h = np.hypot(x, y)
    nz = h > 0
    # (the unit vector is (x, y) / (hypot cos d) with d the float32 estimate's error, |d| < 3e-7: off by d^2 / 2 < 5e-14 -- what rotates the
    #  position loop's setpoint in modes 6 / 7, where 1e-9 would do)
    assert np.abs(cs[nz] - x[nz] / h[nz]).max() < 5e-14 and np.abs(sn[nz] - y[nz] / h[nz]).max() < 5e-14
    nzx = x != 0
    assert np.abs(rc[nzx] * x[nzx] - 1.0).max() < 4e-16
    ax = np.abs(x[nzx])
    assert np.abs(sq[nzx] / np.sqrt(ax) - 1.0).max() < 4e-16
    # asin through atan2(s, sqrt((1 - s)(1 + s))), |s| <= 0.99998 (the gimbal-lock branch takes over at 0.99999)
    s = np.clip(y / np.sqrt(x * x + y * y + 1e-300), -0.99998, 0.99998)
    e_as = np.abs(asn - np.arcsin(s))
    assert e_as.max() < 1e-13, e_as.max()  # (asin's own conditioning at 0.99998: 1 / cos = 160 x the argument's last bit)
    print(f"atan2_d worst {e_at.max():.2e}, unit vector {np.abs(cs[nz] - x[nz] / h[nz]).max():.2e}, rcp_d {np.abs(rc[nzx] * x[nzx] - 1.0).max():.2e}, "
          f"sqrt_pos_d {np.abs(sq[nzx] / np.sqrt(ax) - 1.0).max():.2e}, asin {e_as.max():.2e}")
