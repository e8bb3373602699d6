"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the parameter tables agree with the oracle's, the product refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    return True


def test_library_exports_every_declared_symbol(built):
    from pyflyt_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "pyflyt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pf_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 14
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/pyflyt_amd.h but not exported"
    assert set(_lib.EXPORTS) == declared
    L = _lib.lib()  # also checks struct sizes and the ABI version
    assert L.pf_abi_version() == _lib.PF_ABI_VERSION >= 3


def test_no_cpu_fallback(built):
    import torch

    from pyflyt_amd import PyFlytAmdError, build_params
    from pyflyt_amd.engine import BatchEngine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(PyFlytAmdError):
        BatchEngine(build_params("quadx", "hover"), 64)
    from pyflyt_amd.pz_envs import MAFixedwingDogfightEnv, MAQuadXHoverEnv

    for env_cls in (MAFixedwingDogfightEnv, MAQuadXHoverEnv):  # the PettingZoo facades have no CPU path either
        with pytest.raises(PyFlytAmdError):
            env_cls(device="cuda:0")
    # the C entry point itself refuses as well
    from pyflyt_amd import _lib

    ctx = C.c_void_p()
    P = build_params("quadx", "hover")
    rc = _lib.lib().pf_ctx_create(C.byref(P), 64, 0, 0, C.byref(ctx))
    assert rc != 0 and b"no HIP device" in _lib.lib().pf_last_error(None)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "pyflyt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "uav_oracle.h" not in src, f


def sym(a):
    return np.array([[a[0], a[1], a[2]], [a[1], a[3], a[4]], [a[2], a[4], a[5]]])


@pytest.mark.parametrize("vehicle,env", [("quadx", "hover"), ("quadx", "quadx_waypoints"), ("fixedwing", "fixedwing_waypoints")])
def test_param_tables_agree_with_oracle(vehicle, env):
    """Two independent transcriptions of the reference's YAML/URDF numbers (pyflyt_amd/params.py in
    Python, oracle/uav_oracle.c in C) must agree; the C side is golden-checked against the reference."""
    from oracle import oracle as O
    from pyflyt_amd import build_params

    task = "hover" if env == "hover" else "waypoints"
    P = build_params(vehicle, task)
    R = O.make_params(env)
    f = np.float32
    assert np.isclose(P.inv_mass, 1.0 / R.mass, rtol=1e-6)
    np.testing.assert_allclose(list(P.com), list(R.com), rtol=1e-6, atol=1e-9)
    I_ref = np.array([list(r) for r in R.I_own]) + np.array([list(r) for r in R.I_pa])
    np.testing.assert_allclose(sym(P.I_own) + sym(P.I_pa), I_ref, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(sym(P.I_inv), np.array([list(r) for r in R.I_inv]), rtol=1e-5)
    assert P.n_boxes == R.n_boxes and P.n_motors == R.n_motors and P.n_surf == R.n_surf
    assert P.bound_radius >= f(R.bound_radius)
    for i in range(P.n_motors):
        np.testing.assert_allclose(P.motor_fmax[i], R.thrust_coef[i] * R.max_rpm[i] ** 2, rtol=1e-6)
        np.testing.assert_allclose(P.motor_tmax[i], R.torque_coef[i] * R.max_rpm[i] ** 2, rtol=1e-6)
        np.testing.assert_allclose(P.motor_dt_over_tau[i], R.world.dt / R.motor_tau[i], rtol=1e-6)
        np.testing.assert_allclose(list(P.motor_r[i]), list(R.motor_r[i]), atol=1e-9)
    if vehicle == "quadx":
        np.testing.assert_allclose(list(P.drag_const), list(R.drag_const), rtol=1e-6)
        for k in range(4):
            for name in ("kp", "ki", "kd", "lim"):
                np.testing.assert_allclose(list(getattr(P.pid[k], name)), list(getattr(R.pid[k], name)), rtol=1e-6)
        for k in range(2):
            np.testing.assert_allclose(P.zpid[k].kp[0], R.zpid[k].kp[0]); np.testing.assert_allclose(P.zpid[k].kd[0], R.zpid[k].kd[0])
        np.testing.assert_allclose(np.array([list(r) for r in P.motor_map]), np.array([list(r) for r in R.motor_map]))
    else:
        for i in range(5):
            S, T = P.surf[i], R.surf[i]
            np.testing.assert_allclose(S.Cl_alpha_3D, T.Cl_alpha_3D, rtol=1e-6)
            np.testing.assert_allclose(S.aero_tau_eta, T.aero_tau * T.eta, rtol=1e-6)
            np.testing.assert_allclose(S.half_rho_area, T.half_rho * T.area, rtol=1e-6)
            np.testing.assert_allclose(S.inv_pi_aspect, 1.0 / (np.pi * T.aspect), rtol=1e-6)
            np.testing.assert_allclose([S.alpha_0_base, S.alpha_stall_P_base, S.alpha_stall_N_base],
                                       [T.alpha_0_base, T.alpha_stall_P_base, T.alpha_stall_N_base], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(list(S.r), list(T.r), atol=1e-9)
            np.testing.assert_allclose(list(S.torque), list(T.torque_unit), atol=1e-9)
        assert list(P.assist_ids) == list(R.assist_ids)
    assert P.max_steps == R.max_steps and P.env_step_ratio == R.env_step_ratio and P.settle_steps == R.settle_steps
    assert np.isclose(P.dome, R.dome) and P.num_targets == R.num_targets
    assert np.isclose(P.goal_reach_distance, R.goal_reach_distance) or task == "hover"


def test_build_params_validation():
    from pyflyt_amd import build_params

    with pytest.raises(ValueError, match="agent_hz"):  # quadx_base_env.py:47-52
        build_params("quadx", "hover", agent_hz=50)
    with pytest.raises(ValueError, match="angle_representation"):
        build_params("quadx", "hover", angle_representation="matrix")
    with pytest.raises(ValueError):
        build_params("rocket", "hover")
    P = build_params("quadx", "hover", agent_hz=30, max_duration_seconds=5.0)
    assert P.env_step_ratio == 4 and P.max_steps == 150
    # the contact model's parameters (include/pyflyt_amd.h at pf_params.contact_response): world_options reach the block, and the
    # values the device code cannot take are refused on the host
    P = build_params("quadx", "hover", world_options=dict(contact_report_distance=0.02, contact_break_distance=0.01, contact_manifold_points=8,
                                                          contact_iters=10, contact_residual_threshold=0.0))
    assert (P.contact_manifold_points, P.contact_iters) == (8, 10) and P.contact_residual_threshold == 0.0
    assert abs(P.contact_report_distance - 0.02) < 1e-8 and abs(P.contact_break_distance - 0.01) < 1e-8
    with pytest.raises(ValueError, match="contact_manifold_points"):
        build_params("quadx", "hover", world_options=dict(contact_manifold_points=6))
    with pytest.raises(ValueError, match="non-negative"):
        build_params("quadx", "hover", world_options=dict(contact_slop=-1e-3))


def test_bench_replays_counters_only_for_the_kernels_they_were_collected_on(tmp_path, monkeypatch):
    """bench.py: roofline.traffic / roofline.issue come from the committed PMC collection -- keyed by the source hash of the device
    code, null with the reason when the kernels have changed since, when the env / batch is not covered, or when there is no file."""
    import json

    import bench

    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_record("hover", 65536)[0] is None
    rec = {"source_hash": h, "source": "x", "envs": {"hover": {"batch": 65536, "hbm_bytes_per_launch": 22e6, "valu_per_wave": 2300.0, "salu_per_wave": 360.0,
                                                                 "clocks_per_inst": 8.0}}}
    (prof / "pmc_latest.json").write_text(json.dumps(rec))
    ent, why = bench.pmc_record("hover", 65536)
    assert why is None and ent["hbm_bytes_per_launch"] == 22e6
    r = bench.roofline_block("hover", 65536, 10.5e-6, "k")
    assert r["traffic"] == 22e6 and abs(r["frac"] - 330 * 65536 / 10.5e-6 / 1e9 / 8000.0) < 1e-12
    assert abs(r["issue"]["min_us"] - 2300.0 * 4 / 2400.0) < 1e-9 and abs(r["issue"]["frac"] - r["issue"]["min_us"] / 10.5) < 1e-9
    assert "lone_wave" not in r["issue"]  # (a collection without the issue-slot count: no lone-wave floor is invented)
    rec["envs"]["hover"]["issue_slots_per_wave"] = 2600.0
    (prof / "pmc_latest.json").write_text(json.dumps(rec))
    lw = bench.roofline_block("hover", 65536, 10.5e-6, "k")["issue"]["lone_wave"]
    # every instruction of a lone wave takes four clocks (profiles/r06/lone_wave_issue.txt) + the launch floor of 1 024 one-wave workgroups
    assert abs(lw["issue_us"] - 2600.0 * 4 / 2400.0) < 1e-9 and abs(lw["min_us"] - (lw["issue_us"] + bench.LAUNCH_FLOOR_US)) < 1e-12
    assert abs(lw["frac"] - lw["min_us"] / 10.5) < 1e-9 and lw["min_us"] < 10.5
    assert bench.pmc_record("hover", 4096)[0] is None and "cover" in bench.pmc_record("quadx_waypoints", 65536)[1]
    rec["source_hash"] = "0" * 16
    (prof / "pmc_latest.json").write_text(json.dumps(rec))
    ent, why = bench.pmc_record("hover", 65536)
    assert ent is None and why.startswith("stale")
    r = bench.roofline_block("hover", 65536, 10.5e-6, "k")
    assert r["traffic"] is None and "issue" not in r and r["traffic_source"].startswith("stale")


def test_quadk_hot_path_selection(built):
    """The specialised kernel is chosen exactly for the configurations it implements."""
    src = open(os.path.join(ROOT, "pyflyt_amd", "csrc", "quadx_fast.hpp")).read()
    assert "P.flight_mode != 0" in src and "PF_TASK_HOVER" in src


def test_integration_md_stub_matches_the_header():
    """INTEGRATION.md section 3 shows the ctypes stub a maintainer of the reference would add: its pf_buffers mirror and the ABI version it
    asserts must be the header's."""
    import re

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    h = open(os.path.join(root, "include", "pyflyt_amd.h")).read()
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    body = re.sub(r"/\*.*?\*/", "", re.search(r"typedef struct pf_buffers \{(.*?)\} pf_buffers;", h, re.S).group(1), flags=re.S)
    header_fields = re.findall(r"\*\s*(\w+)\s*;", body)
    stub = re.search(r"class PfBuffers\(C\.Structure\):.*?for n in \((.*?)\)\]", md, re.S).group(1)
    assert re.findall(r'"(\w+)"', stub) == header_fields
    abi = int(re.search(r"#define PF_ABI_VERSION (\d+)", h).group(1))
    assert [int(v) for v in re.findall(r"pf_abi_version\(\)`? ==? (\d+)", md)] == [abi, abi]


def test_bench_gpus_n_never_runs_as_one_process():
    """`python bench.py --gpus 8` without a torchrun environment must start eight ranks or refuse (round 5's read WORLD_SIZE,
    found 1 and reported a one-GPU line under the eight-GPU command). Here there is no device at all: it refuses, non-zero, no line."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PF_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in r.stderr and "8-GPU label" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a torchrun environment that disagrees with --gpus is an error as well, not a silent relabel
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
