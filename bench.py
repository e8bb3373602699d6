#!/usr/bin/env python3
"""bench.py -- headline benchmark: env-steps/sec of the fused QuadX-Hover env step.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 one rank per GPU -- either launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or, when no torchrun environment is present,
bench.py starts those N ranks ITSELF (the same torch.distributed.run command line, 127.0.0.1 rendezvous) and exits non-zero when
fewer than N devices are visible: `--gpus 8` never reports a one-GPU number. Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric): PyFlyt/QuadX-Hover (flight mode 0, 40 Hz agent -> 6 physics ticks
and 3 control ticks per env step, quaternion observation, dense reward), batch 65 536 drones PER
GPU (weak scaling, config[4]: 524 288 over 8 GPUs), motor noise ON (counter-based Philox,
xi ~ N(4,1) as in motors.py:134-138), uniformly random actions in the action box, NEXT_STEP
auto-reset (gymnasium VectorEnv default) with its 20 settle ticks inside the timed region.
One "step" = one pf_env_step launch over the whole per-GPU batch: actions [n,4] read from HBM,
obs [n,21] / reward / terminated / truncated written to HBM, persistent state round-trips HBM.
The batch shards embarrassingly: no collective in the timed loop.

The K timed steps are replayed from HIP graphs (launch-bound inner loop -> hipGraph), each graph
node one env step reading its own pre-generated action batch (inputs resident in HBM).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SHADER_CLOCK_GHZ = 2.4   # MI355X peak engine clock (same guide); the phase traces read 2.2-2.25 GHz under this load
LAUNCH_FLOOR_US = 1.42    # a graph launch of 1 024 one-wave workgroups beyond its waves' life (profiles/r05/ubench_dispatch_ramp.txt)
# algorithmic HBM bytes per env step per lane:
#   reads  128 B = 7 state float4 groups (112) + action float4 (16)
#   writes 202 B = 7 state groups (112) + obs 21 f32 (84) + reward (4) + terminated (1) + truncated (1)
ALGO_BYTES = {"hover": 330, "quadx_waypoints": 442, "fixedwing_waypoints": 418,
              # dogfight (team_size 2): 15 state groups read + written (480), action (16), obs 65 f32 (260), reward + flags (6)
              "dogfight": 762,
              # PettingZoo MA-Hover, one world per 4 agents: 11 state groups read + written (352), action (16), obs 24 f32 (96), reward + flags (6)
              "ma_hover": 470}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=65536, help="lanes per GPU")
    ap.add_argument("--env", default="hover", choices=["hover", "quadx_waypoints", "fixedwing_waypoints", "dogfight", "ma_hover"])
    ap.add_argument("--noise", default="philox", choices=["philox", "off"])
    ap.add_argument("--graph-steps", type=int, default=100, help="env steps captured per HIP graph")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-min-steps", type=int, default=200,
                    help="below this many timed steps the launches are issued one by one through prepared steps (BatchEngine.prepare_step) "
                         "instead of a HIP graph: launching an instantiated graph costs ~0.1 ms of latency, which 2 000 steps amortise and "
                         "20 steps (the driver's short run) do not")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch lanes PER GPU (BASELINE config 5); strong: --batch lanes in TOTAL cut over the GPUs "
                         "(the metric's 'batch=65536 at 1/2/4/8' point)")
    ap.add_argument("--no-contact-response", action="store_true",
                    help="opt OUT of stepSimulation's contact solve (world_options contact_response=False): contact DETECTION only, "
                         "a departure from the reference; diagnostic A/B, never the headline configuration (the default has the solve on)")
    ap.add_argument("--world", action="append", default=[], metavar="KEY=VALUE",
                    help="override an entry of pyflyt_amd.params.WORLD (diagnostic), e.g. --world contact_iters=6")
    ap.add_argument("--ring", type=int, default=100, help="least number of entries of the action ring (independent uniform draws per lane and entry)")
    ap.add_argument("--seed", type=int, default=0, help="Philox seed of the env's noise, spawn and action draws")
    ap.add_argument("--preroll", type=int, default=400,
                    help="env steps run as part of the SETUP, before the W warm-up steps (one pf_rollout launch, or eager steps where there is "
                         "no rollout): after the initial reset every lane is in the same phase of its first episode -- with random actions the "
                         "whole batch reaches the floor (and the contact solve) within the same few steps about half a second in -- and "
                         "the figure is for the steady state of a long random-action rollout, whatever K and W are")
    ap.add_argument("--dogfight-actions", default="gentle", choices=["gentle", "uniform"],
                    help="dogfight env: gentle commands around level flight (everybody stays airborne) or the action box's uniform "
                         "distribution (aircraft reach the ground within seconds: the contact-solve regime)")
    ap.add_argument("--flight-mode", type=int, default=0, help="QuadX flight mode -1..7 (auxiliary figures; the metric is quoted on mode 0)")
    ap.add_argument("--rollout-steps", type=int, default=100, help="env steps per pf_rollout launch of the second, state-resident figure (0 = skip)")
    ap.add_argument("--min-timed-ms", type=float, default=25.0,
                    help="floor on the timed region: the K steps are repeated (whole multiples of K, back to back, no host synchronisation in "
                         "between) until the region lasts at least this long, and every figure stays per step. K = 20 steps of 11 us are "
                         "0.2 ms: on one GPU that measures the launch and completion latency of the run as much as the kernels, and on eight "
                         "the max over ranks of such a region measures launch skew. 0 = time exactly K steps. (25 ms since round 6: at 5 ms the one "
                         "launch-to-first-kernel latency of the region, ~0.2 ms, was still 3 % of it -- 9.59 us per step by the wall clock against "
                         "9.27 by the events on the same run)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the secondary BASELINE configs (Hover 4 096, QuadX-Waypoints 65 536, Fixedwing-Waypoints 65 536) that the "
                         "default single-GPU hover run times after the headline and reports under `configs`")
    ap.add_argument("--no-facade", action="store_true",
                    help="skip the `facade` block: the drop-in surface itself (gym_envs.make_vec(...).step(), pz_envs.MAQuadXHoverEnv.step()) in a "
                         "closed loop, eager and captured in a HIP graph")
    ap.add_argument("--config-steps", type=int, default=2000,
                    help="timed steps per secondary config (QuadX-Waypoints has a heavy tail -- the launches in which a lane solves a floor contact --: "
                         "500 steps were one deterministic sample 1.7 us above the 2000-step mean)")
    return ap.parse_args()


def self_launch(n_gpus):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks here (the command line the contract
    names), pass their output through, exit with their status. Refuses when fewer than N devices are visible (PF_BENCH_SINGLE_DEVICE=1,
    the one-GPU launcher test, puts every rank on cuda:0)."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("PF_BENCH_SINGLE_DEVICE") != "1" and have < n_gpus:
        raise SystemExit(f"bench.py --gpus {n_gpus}: {have} ROCm device(s) visible to this process; refusing to report a {have}-GPU figure under an "
                         f"{n_gpus}-GPU label (one rank per GPU: no oversubscription, no CPU fallback)")
    if have == 0:
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback exists for the product path)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PF_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL's intra-node transport needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def make_engine(env, batch, device, lane_offset, noise, contact_response=True, world=(), flight_mode=0, seed=0):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    if env == "dogfight":  # MAFixedwingDogfightEnv defaults: 2 v 2 Acrowing per world (4 adjacent lanes), world_scale 5; no auto-reset
        # in the PettingZoo API -- finished agents are culled and their aircraft fly on, the per-step work does not change
        P = build_params("fixedwing", "dogfight", noise=noise, autoreset="off", seed=0, angle_representation="euler",
                         vehicle_options=dict(drone_model="acrowing"), world_options=dict(world_scale=5.0))
        return BatchEngine(P, batch, device=device, lane_offset=lane_offset)
    if env == "ma_hover":  # MAQuadXHoverEnv, four agents per (shared) world 2 m apart, no auto-reset in the PettingZoo API
        import numpy as np
        import torch

        from pyflyt_amd.params import quat_from_euler
        sp = np.array([[-1.0, -1.0, 1.0], [1.0, -1.0, 1.0], [-1.0, 1.0, 1.0], [1.0, 1.0, 1.0]])
        P = build_params("quadx", "ma_hover", noise=noise, autoreset="off", seed=0, agents_per_world=4, start_pos=sp[0], flight_mode=flight_mode,
                         world_options=dict(contact_response=True))
        eng = BatchEngine(P, batch, device=device, lane_offset=lane_offset)
        side = np.zeros((batch, 12), dtype=np.float32)
        side[:, :7] = np.tile(np.concatenate([sp, np.tile(quat_from_euler((0, 0, 0)), (4, 1))], axis=1), (batch // 4, 1))
        eng.state[12:15] = torch.tensor(side, device=device).view(batch, 3, 4).permute(1, 0, 2)
        return eng
    vehicle, task = {"hover": ("quadx", "hover"), "quadx_waypoints": ("quadx", "waypoints"),
                     "fixedwing_waypoints": ("fixedwing", "waypoints")}[env]
    wo = {} if contact_response else dict(contact_response=False)  # (default: the solve is ON, as in the reference)
    for kv in world:
        k, v = kv.split("=", 1)
        wo[k] = float(v) if "." in v or "e" in v.lower() else int(v)
    kw = dict(flight_mode=flight_mode) if vehicle == "quadx" else {}
    P = build_params(vehicle, task, noise=noise, autoreset="next_step", seed=seed, world_options=wo or None, **kw)
    return BatchEngine(P, batch, device=device, lane_offset=lane_offset)


def cpu_baseline(env, noise, seconds):
    """The fp64 oracle (a port, not the reference itself) on the host cores, OpenMP over lanes,
    same tick structure and auto-reset, bounded to ~`seconds` of wall time."""
    import numpy as np

    from oracle import oracle as O

    n = 4096
    mode = O.NOISE_PHILOX if noise == "philox" else O.NOISE_OFF
    P = O.make_params(env, noise_mode=mode, seed=0)
    ob = O.OracleBatch(P, n)
    ob.reset()
    rng = np.random.default_rng(0)
    low = np.array([-np.pi] * 3 + [0.0]) if env != "fixedwing_waypoints" else -np.ones(4)
    high = np.array([np.pi] * 3 + [0.8]) if env != "fixedwing_waypoints" else np.ones(4)
    # the same action process as the GPU leg: a ring of independent uniform draws per lane and entry, never shorter than 100 entries
    # (a ring that repeats within an episode's length is a different workload: see --ring)
    R = 100
    acts = [rng.uniform(low, high, size=(n, 4)).astype(np.float32) for _ in range(R)]
    for j in range(R):  # (the episode phases decorrelate before anything is timed, as the GPU leg's preroll does)
        ob.step(acts[j], autoreset=1)
    # three samples of a third of the budget each: the host is shared, and one sample swung 0.6-2.8 M with the neighbours' load
    samples, k = [], 0
    for _ in range(3):
        t0 = time.perf_counter()
        k0 = k
        while time.perf_counter() - t0 < seconds / 3.0:
            ob.step(acts[k % R], autoreset=1)
            k += 1
        samples.append(n * (k - k0) / (time.perf_counter() - t0))
    samples.sort()
    cores = O.lib().orc_num_threads()
    # BASELINE.md B1 (config 1 plumbing): one env, 1000 random-action steps with reset on term/trunc, one core
    one = O.OracleBatch(P, 1)
    one.reset()
    t1 = time.perf_counter()
    for j in range(1000):
        one.step(acts[j % R][:1], autoreset=1)
    dt1 = time.perf_counter() - t1
    return {"value": samples[1], "min": samples[0], "max": samples[2], "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"fp64 C restatement (oracle/uav_oracle.c), {env}, batch {n}, action ring of {R} entries, {k} steps in three samples of {seconds / 3.0:.1f} s "
                      "(value = their median), OpenMP over lanes",
            "single_env_1core": {"value": 1000 / dt1, "unit": "env-steps/s", "cores": 1,
                                 "sample": "1 env, 1000 steps, NEXT_STEP auto-reset, incl. ctypes call overhead per step"}}


def preroll(eng, ring, steps, step_index0):
    """`steps` untimed env steps: one pf_rollout launch; eager steps only where the library says the task has no rollout."""
    from pyflyt_amd import _lib as PL

    try:
        eng.rollout(steps, step_index0=step_index0)
    except PL.PfError as e:
        if e.code != PL.ERR_UNSUPPORTED:
            raise
        for i in range(steps):
            eng.env_step(ring[i % len(ring)])


def source_hash():
    """What the device code is built from (sources, the assembly repair, the compiler flags): the key that ties a committed PMC
    collection to the kernels it was collected on."""
    import hashlib

    import __graft_entry__ as G

    h = hashlib.sha256()
    for f in G.HIP_DEPS:
        if f.endswith(("__graft_entry__.py",)):
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(G.HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def pmc_record(env, n):
    """The committed PMC collection for this env / batch (profiles/pmc_latest.json), or (None, why): rocprofv3 --pmc cannot run inside
    the timed bench, so HBM traffic and the instruction counts are REPLAYED from it -- only while the kernels are the ones it was
    collected on (source_hash)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(pmc):
        return None, "no profiles/pmc_latest.json"
    try:
        rec = json.load(open(pmc))
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable profiles/pmc_latest.json: {e}"
    ent = rec.get("envs", {}).get(env)
    if ent is None or ent.get("batch") != n:
        return None, f"the collection does not cover {env} at batch {n}"
    if rec.get("source_hash") != source_hash():
        return None, f"stale: collected on kernels {rec.get('source_hash')}, these are {source_hash()}"
    ent = dict(ent)
    ent["source"] = rec.get("source", "profiles/pmc_latest.json")
    return ent, None


def roofline_block(env, n, per_launch_s, kernel):
    """HBM roofline of one launch shape + the second, binding one: VALU issue (one wave per SIMD, a wave64 VALU instruction
    occupies its SIMD for four clocks: valu_per_wave x 4 / clock is the least a wave's life can be)."""
    algo = ALGO_BYTES[env] * n
    achieved = algo / per_launch_s / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "traffic_source": None, "kernel": kernel, "algorithmic_bytes_per_launch": algo, "launch_us": per_launch_s * 1e6}
    ent, why = pmc_record(env, n)
    if ent is None:
        out["traffic_source"] = why
        return out
    out["traffic"] = ent["hbm_bytes_per_launch"]
    out["traffic_source"] = "replayed from " + ent["source"]
    if "valu_per_wave" in ent:
        clk = SHADER_CLOCK_GHZ
        min_us = ent["valu_per_wave"] * 4.0 / (clk * 1e3)
        out["issue"] = {"bound": "valu-issue", "valu_per_wave": ent["valu_per_wave"], "salu_per_wave": ent.get("salu_per_wave"),
                        "clocks_per_inst": ent.get("clocks_per_inst"), "clock_ghz": clk, "min_us": min_us, "frac": min_us / (per_launch_s * 1e6),
                        "note": "one wave per SIMD: a wave64 VALU instruction holds its SIMD for 4 clocks; clocks_per_inst = measured wave "
                                "cycles / (VALU + SALU) of the same collection"}
        if ent.get("issue_slots_per_wave"):
            # the tighter floor of a LONE wave (profiles/r06/lone_wave_issue.txt, measured on MI355X): it issues ONE instruction of any
            # kind -- vector, scalar, LDS, memory, wait, branch -- per 4 clocks (5 in a run of 8-byte encodings: 1.6 B of code per clock),
            # dependent or not, nothing co-issues, a transcendental takes two slots; and a launch of 1 024 one-wave workgroups costs
            # 1.42 us beyond its slowest wave's life (profiles/r05/ubench_dispatch_ramp.txt)
            slots_us = ent["issue_slots_per_wave"] * 4.0 / (clk * 1e3)
            out["issue"]["lone_wave"] = {"issue_slots_per_wave": ent["issue_slots_per_wave"], "issue_us": slots_us, "launch_floor_us": LAUNCH_FLOOR_US,
                                         "min_us": slots_us + LAUNCH_FLOOR_US, "frac": (slots_us + LAUNCH_FLOOR_US) / (per_launch_s * 1e6)}
    return out


KERNEL_OF = {"fixedwing_waypoints": "pf::fixedwing_wp_env_kernel", "dogfight": "pf::dogfight_env_kernel", "ma_hover": "pf::quadx_m0_env_kernel<MA_HOVER, .., SHARED>"}


def time_config(env, batch, device, args):
    """One secondary BASELINE config, the headline's method in small: engine, 100-entry action ring, reset, 400 steps of episode-phase
    preroll, a HIP graph of 100 steps replayed once (upload), then `--config-steps` steps timed with HIP events on the launch stream."""
    import torch

    eng = make_engine(env, batch, device, lane_offset=0, noise=args.noise, seed=args.seed)
    ring = [torch.empty(batch, 4, dtype=torch.float32, device=device) for _ in range(100)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    preroll(eng, ring, 400, 1 << 20)
    eng.env_step(ring[0])
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=device)
    g = 100
    reps = max(1, args.config_steps // g)
    with torch.cuda.stream(stream):
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for i in range(g):
                eng.env_step(ring[i])
        graph.replay()
        preroll(eng, ring, 300, (1 << 20) + 400)  # (clock spin-up right in front of the timed replays)
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            graph.replay()
        e1.record(stream)
        stream.synchronize()
    per = e0.elapsed_time(e1) * 1e-3 / (reps * g)
    assert torch.isfinite(eng.obs).all(), f"non-finite observation in config {env}"
    r = roofline_block(env, batch, per, KERNEL_OF.get(env, "pf::quadx_m0_env_kernel"))
    out = {"workload": f"{env}, batch {batch}, random actions, motor noise {args.noise}, NEXT_STEP auto-reset, contact response on", "steps": reps * g,
           "launch_us": per * 1e6, "value": batch / per, "unit": "env-steps/s", "roofline": r}
    if args.rollout_steps > 0:
        # the state-resident figure (pf_rollout, K steps per launch): the only way a small batch leaves the per-launch floor --
        # 4 096 lanes are 64 waves on 1 024 SIMDs, a launch per step costs its fixed 8 us whatever the kernel does
        kk = args.rollout_steps
        with torch.cuda.stream(stream):
            eng.rollout(kk, step_index0=0)
            stream.synchronize()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            r0.record(stream)
            for i in range(5):
                eng.rollout(kk, step_index0=(i + 1) * kk)
            r1.record(stream)
            stream.synchronize()
        rper = r0.elapsed_time(r1) * 1e-3 / (5 * kk)
        assert torch.isfinite(eng._traj["obs"]).all(), f"non-finite observation in the rollout of config {env}"
        out["rollout"] = {"k": kk, "launches": 5, "us_per_step": rper * 1e6, "value": batch / rper, "unit": "env-steps/s",
                          "note": "pf_rollout: k env steps per launch, state in registers, on-device action sampling; bit-identical to k x pf_env_step"}
    return out


def time_config_guarded(env, batch, device, args):
    """A secondary config must not take the headline line down with it: an exception is recorded in its place."""
    try:
        return time_config(env, batch, device, args)
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def _loop_us(fn, steps, torch):
    """Wall-clock microseconds per call of `fn(i)` over `steps` calls, device drained before and after (the eager closed loop)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / steps


def _graph_us(body, g, reps, stream, torch):
    """`body(i)` for i < g captured in ONE HIP graph on `stream`, replayed once (upload), then `reps` replays timed with HIP events
    on that stream: microseconds per captured step."""
    with torch.cuda.stream(stream):
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for i in range(g):
                body(i)
        graph.replay()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            graph.replay()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * g), graph


def facade_block(device, args, ms_per_step):
    """The drop-in surface itself, measured: what a user of `make_vec("PyFlyt/QuadX-Hover-v4", 65536).step(actions)` and of
    `MAQuadXHoverEnv(num_envs=16384).step(actions)` gets per step -- (a) `step_only`: env.step() over the headline's own action ring,
    eager (wall clock, host included) and captured in a HIP graph (events): the facade's cost over the bare pf_env_step launch of the
    headline; (b) `closed_loop`: a trivial on-device policy actions = clamp(obs @ W + b) writing into a fixed action buffer and
    env.step() on it, eager and captured, with the policy's two torch kernels alone next to it. Nothing here reads `infos`."""
    import torch

    from pyflyt_amd.gym_envs import make_vec
    from pyflyt_amd.pz_envs import MAQuadXHoverEnv

    steps, g = 2000, 100
    stream = torch.cuda.Stream(device=device)

    def measure(step, obs, act, ring, policy_wb, lo, hi):
        W, b = policy_wb
        out = {}
        for i in range(max(200, 2 * len(ring))):  # (every ring entry's buffer block prepared, clocks up)
            step(ring[i % len(ring)])
        eager = _loop_us(lambda i: step(ring[i % len(ring)]), steps, torch)
        gus, gr = _graph_us(lambda i: step(ring[i % len(ring)]), g, steps // g, stream, torch)
        out["step_only"] = {"eager_us_per_step": eager, "graph_us_per_step": gus, "graph_vs_headline_launch": gus / (ms_per_step * 1e3),
                            "what": f"env.step(ring[i]) over a {len(ring)}-entry ring of uniform action draws, {steps} steps; infos not read"}
        del gr

        def policy():
            torch.addmm(b, obs, W, out=act)
            torch.clamp(act, min=lo, max=hi, out=act)

        def closed(i):
            policy()
            step(act)

        for i in range(200):
            closed(i)
        eager_c = _loop_us(closed, steps, torch)
        gus_c, gr = _graph_us(closed, g, steps // g, stream, torch)
        del gr
        pol, gr = _graph_us(lambda i: policy(), g, steps // g, stream, torch)
        del gr
        out["closed_loop"] = {"eager_us_per_step": eager_c, "graph_us_per_step": gus_c, "policy_graph_us_per_step": pol,
                              "graph_minus_policy_us": gus_c - pol,
                              "policy": "actions = clamp(obs @ W + b) into a fixed action tensor: torch.addmm + torch.clamp (two kernels), then env.step(actions)"}
        return out

    res = {"what": "the Gymnasium-VectorEnv / PettingZoo-shaped step() of the package in a closed loop, eager (wall clock) and captured in a HIP graph "
                   "(HIP events); per step: one foreign call, no torch kernel, no host synchronisation (tests/test_gpu_api.py)"}
    # ---- gym_envs.make_vec("PyFlyt/QuadX-Hover-v4", 65536)
    n = 65536
    env = make_vec("PyFlyt/QuadX-Hover-v4", n, device=device, seed=args.seed)
    eng = env.engine
    ring = [torch.empty(n, 4, dtype=torch.float32, device=device) for _ in range(100)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    obs, _ = env.reset()
    preroll(eng, ring, 400, 1 << 20)  # (episode phases decorrelated, as the headline's setup does)
    lo = torch.tensor(env.single_action_space.low, device=device)
    hi = torch.tensor(env.single_action_space.high, device=device)
    # a hovering linear policy: rate commands against the attitude (quaternion x, y at obs[3], obs[4]) and the body rates (obs[0:3]),
    # thrust around the hover command against height and climb rate (obs[12], obs[9]): episodes run to truncation
    W = torch.zeros(eng.obs_dim, 4, device=device)
    W[0, 0] = W[1, 1] = W[2, 2] = -0.2
    W[3, 0] = W[4, 1] = -4.0
    W[12, 3], W[9, 3] = -0.3, -0.2
    b = torch.tensor([0.0, 0.0, 0.0, 0.3772 + 0.3], device=device)
    act = torch.zeros(n, 4, device=device)
    v = measure(lambda a: env.step(a), obs, act, ring, (W, b), lo, hi)
    v["id"], v["num_envs"] = "PyFlyt/QuadX-Hover-v4", n
    assert torch.isfinite(eng.obs).all(), "non-finite observation in the facade loop"
    res["vector_env"] = v
    env.close()
    del env, eng, ring, act
    # ---- pz_envs.MAQuadXHoverEnv, 16 384 copies of the four-agent env (65 536 lanes), every agent its own lane (the default)
    E = 16384
    ma = MAQuadXHoverEnv(num_envs=E, device=device, seed=args.seed, cull_agents=False)
    ma.reset()
    eng = ma.engine
    bufs = ma.action_buffers()
    flat_act = ma._act_flat  # the env's own [E * A, 4] action tensor, the storage behind action_buffers()
    # step_only: small rate commands, thrust around the hover command, written into the env's action tensor once and held
    eng.sample_actions(flat_act, 0)
    flat_act[:, :3].mul_(0.01 / 3.14159265)
    flat_act[:, 3].mul_(0.001).add_(0.3772)
    lo = torch.tensor(ma.action_space().low, device=device)
    hi = torch.tensor(ma.action_space().high, device=device)
    W = torch.zeros(eng.obs_dim, 4, device=device)
    W[0, 0] = W[1, 1] = W[2, 2] = -0.2
    W[3, 0] = W[4, 1] = -4.0
    W[12, 3], W[9, 3] = -0.3, -0.2
    b = torch.tensor([0.0, 0.0, 0.0, 0.3772 + 0.3], device=device)

    m = measure(lambda a: ma.step(bufs), eng.obs, flat_act, [flat_act], (W, b), lo, hi)
    m["step_only"]["what"] = f"env.step(env.action_buffers()) with held hover commands in the env's own action tensor, {steps} steps; infos not read"
    m["env"], m["num_envs"], m["agents"], m["lanes"] = "MAQuadXHoverEnv(cull_agents=False)", E, 4, 4 * E
    assert torch.isfinite(eng.obs).all(), "non-finite observation in the PettingZoo facade loop"
    res["pettingzoo"] = m
    ma.close()
    res["eager_us_per_step"] = res["vector_env"]["step_only"]["eager_us_per_step"]
    res["graph_us_per_step"] = res["vector_env"]["step_only"]["graph_us_per_step"]
    res["headline_launch_us"] = ms_per_step * 1e3
    return res


def main():
    args = parse()
    import torch

    from pyflyt_amd.dist import env_rank_world, strong_shard, weak_shard

    rank, local_rank, world = env_rank_world()
    if world != args.gpus:
        if "WORLD_SIZE" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        self_launch(args.gpus)  # (does not return: the N ranks' exit status is this process's)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback exists for the product path)")
    # PF_BENCH_SINGLE_DEVICE=1 + PF_BENCH_BACKEND=gloo: every rank on cuda:0, host-side collectives -- the N > 1 launcher path
    # (torchrun env, sharding, barrier, max over ranks, rank-0 line) on a one-GPU box (tests/test_gpu_dist_launch.py); RCCL itself
    # refuses two ranks on one device
    dev_index = 0 if os.environ.get("PF_BENCH_SINGLE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = os.environ.get("PF_BENCH_BACKEND", "nccl")
    if world > 1 or os.environ.get("PF_BENCH_FORCE_DIST") == "1":  # (the env switch exercises the RCCL path on a 1-GPU box)
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    red_dev = device if backend == "nccl" else torch.device("cpu")  # where the clocks are max-reduced

    # per-GPU slice; no collective in the timed loop
    shard = weak_shard(args.batch, rank, world) if args.scaling == "weak" else strong_shard(args.batch, rank, world, unit=4 if args.env in ("dogfight", "ma_hover") else 1)
    n = shard.lanes
    eng = make_engine(args.env, n, device, lane_offset=shard.lane_offset, noise=args.noise, contact_response=not args.no_contact_response, world=args.world, flight_mode=args.flight_mode, seed=args.seed)
    # steps per HIP graph: a graph is replayed whole, so it is no longer than the timed run -- nor than the warm-up, so that the
    # warm-up can include one replay of it (the first launch of a freshly instantiated graph carries its upload: with the
    # driver's --steps 20 --warmup 5 that one-off cost was a third of the timed region)
    # steps per HIP graph: a graph is replayed whole, so it is no longer than the timed run
    g = max(1, min(args.graph_steps, args.steps))
    # --min-timed-ms: how many times the K steps are repeated (from the algorithmic floor, 8 us per step at 65 536 lanes: the count
    # only has to get the region over the floor)
    repeats = 1
    if args.min_timed_ms > 0:
        est_ms = args.steps * 8e-3 * max(1.0, n / 65536.0)
        repeats = max(1, int(-(-args.min_timed_ms // est_ms)))
    if repeats > 1 and not args.no_graph:
        # the repeats replay one graph: it must not be shorter than the action ring's least length (a 20-step graph replayed is a
        # 20-entry ring: see --ring), so it holds whole multiples of K up to that length, and the region whole replays of it
        per_graph = -(-args.ring // args.steps)
        g = args.steps * per_graph
        repeats = per_graph * (-(-repeats // per_graph))
    # the action ring: independent uniform draws per lane and entry. It is never shorter than --ring (default 100) entries, however
    # short the run: a ring that repeats within an episode's length is a different action process -- every lane keeps a thrust
    # bias, the drones sink, and the floor's contact solve runs in every launch (profiles/tools/solver_trace.py WHAT=rates:
    # 6 solves per launch with a ring of 16, 1.3 with 20, 0.03 with 100, 0.04 with a fresh draw every step)
    R = max(g, min(args.ring, max(1, (2 << 30) // (16 * n))))  # (at most 2 GiB of actions)
    ring = [torch.empty(n, 4, dtype=torch.float32, device=device) for _ in range(R)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
        if args.env == "ma_hover":  # small rate commands, thrust just above the hover value: everybody stays airborne
            a[:, :3].mul_(0.01 / 3.14159265)
            a[:, 3].mul_(0.001).add_(0.3772)  # (the hover command, bisected with the oracle: 0.3770 holds z over 100 steps; +-0.001 drifts < 1 m over the run)
        if args.env == "dogfight" and args.dogfight_actions == "gentle":
            # uniform actions over the whole box fly every aircraft into the ground within seconds, and a world of wrecks at rest
            # on the floor (contact solve every tick for every lane) is not the regime a policy trains in: gentle commands
            # around level flight instead (stick +-0.15, throttle command 0.25..0.55)
            a.mul_(0.15)
            a[:, 3] += 0.4
    eng.env_reset()
    # ---- setup (untimed, not the warm-up; reported in config.setup): (1) decorrelate the lanes' episode phases (see --preroll);
    # (2) run the step kernel once so that its code object is loaded before the capture; (3) instantiate the graph and replay it
    # ONCE: the first launch of a freshly instantiated graph carries its upload to the device -- with the driver's
    # `--steps 20 --warmup 5` that one-off cost was a third of the timed region, and the warm-up (5 steps) is shorter than the
    # graph (20), so it cannot absorb it.
    setup_steps = 0
    if args.preroll > 0 and args.env not in ("dogfight", "ma_hover"):
        preroll(eng, ring, args.preroll, 1 << 20)
        setup_steps += args.preroll
    eng.env_step(ring[0])
    setup_steps += 1
    torch.cuda.synchronize()

    stream = torch.cuda.Stream(device=device)
    graph = None
    with torch.cuda.stream(stream):
        # (a region made of repeats is long enough to amortise a graph launch whatever K is: one graph of K steps, replayed)
        use_graph = not args.no_graph and (args.steps >= args.graph_min_steps or repeats > 1)
        launchers = None
        if not use_graph and not args.no_graph and args.noise != "inject":
            import ctypes

            launchers = [eng.prepare_step(a) for a in ring]
            sp = ctypes.c_void_p(stream.cuda_stream)
        if use_graph:
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for i in range(g):
                    eng.env_step(ring[i])
            graph.replay()
            setup_steps += g
            stream.synchronize()

        pos = 0  # (position in the ring of the next eagerly launched step: the timed steps go on where the warm-up stopped)

        def run(k):
            nonlocal pos
            done = 0
            if graph is not None:
                while k - done >= g:
                    graph.replay()
                    done += g
            if launchers is not None:
                while done < k:
                    launchers[pos % R](sp)
                    done += 1; pos += 1
            while done < k:
                eng.env_step(ring[pos % R])
                done += 1; pos += 1

        # clock spin-up (setup): the host-side preparation above left the GPU idle for milliseconds, and a short run (the driver's
        # 5 + 20 steps are 0.3 ms of work) would be measured on a clock that is still ramping: keep the device busy right up to
        # the warm-up, on the same stream, with no host synchronisation in between
        if args.preroll > 0 and args.env not in ("dogfight", "ma_hover"):
            preroll(eng, ring, min(args.preroll, 300), (1 << 20) + args.preroll)
            setup_steps += min(args.preroll, 300)
        run(args.warmup)  # the W warm-up steps (whole graph replays first, the remainder eagerly)
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        solver_trace = None
        if os.environ.get("PF_BENCH_SOLVER_TRACE"):  # (diagnostic, with the -DPF_PHASE_TRACE variant library: contact-solver calls inside the timed region)
            import ctypes as _C

            solver_trace = (_C.c_ulonglong * 8)()
            eng.lib.pf_debug_solver_trace(solver_trace)  # (reads and clears)
        # --min-timed-ms: the K steps repeated `repeats` times back to back (estimate from the warm-up's own kernels: two timed
        # replays of K steps would do, but an estimate from the algorithmic floor is enough to pick the count -- 8 us per step)
        ev0.record(stream)  # (in front of the wall clock: the event pair brackets a superset of the timed region)
        t0 = time.perf_counter()
        run(repeats * args.steps)  # (whole graph replays when the region is made of repeats)
        ev1.record(stream)
        while not ev1.query():  # (poll first: a blocking synchronize sleeps on an interrupt, tens of microseconds on a 0.25 ms run)
            pass
        torch.cuda.synchronize()  # (every stream of the device, the launch stream included)
        wall = time.perf_counter() - t0
        if solver_trace is not None:
            eng.lib.pf_debug_solver_trace(solver_trace)
            print(f"[solver trace] timed region: {solver_trace[0]} solver calls, {solver_trace[5]} sweeps, {solver_trace[4]} lanes with contacts", file=sys.stderr)
        if dist is not None:
            dist.barrier()
    ev_ms = ev0.elapsed_time(ev1)
    timed_steps = repeats * args.steps
    t = torch.tensor([wall, ev_ms * 1e-3], dtype=torch.float64, device=red_dev)
    per_rank = None
    if dist is not None:
        # every rank's own figures next to the max over ranks: [wall, events] per rank (a slow rank or launch skew shows here)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [[float(x[0]), float(x[1])] for x in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, ev_max = float(t[0]), float(t[1])

    # sanity: the simulation actually advanced and stayed finite
    ints = eng.ints()
    assert torch.isfinite(eng.obs).all(), "non-finite observation"
    assert int(ints[:, 2].min()) >= timed_steps, "event counter did not advance"  # (counts env steps and resets since context creation)
    from pyflyt_amd import _lib as PL

    nonfinite = int(((ints[:, 1] & PL.F_NONFINITE) != 0).sum())  # lanes whose NaN/Inf guard bit is up at the end

    # second figure: the same env steps, K per launch, with the lane state resident in registers (pf_rollout:
    # actions sampled on device with pf_sample_actions' keys, every step's obs / action / reward / flags written
    # to trajectory buffers). Timed with HIP events on the launch stream; same barrier / max-over-ranks rule.
    roll = None
    if args.rollout_steps > 0 and args.env not in ("dogfight", "ma_hover"):
        kk = args.rollout_steps
        reps = max(5, args.steps // kk)  # (at least five launches: a single one is dominated by its launch / first-touch overheads)
        with torch.cuda.stream(stream):
            eng.rollout(kk, step_index0=0)  # warm-up launch (allocates the trajectory buffers)
            stream.synchronize()
            if dist is not None:
                dist.barrier()
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tr0 = time.perf_counter()
            r0.record(stream)
            for i in range(reps):
                eng.rollout(kk, step_index0=(i + 1) * kk)
            r1.record(stream)
            stream.synchronize()
            rwall = time.perf_counter() - tr0
            if dist is not None:
                dist.barrier()
        rt = torch.tensor([rwall, r0.elapsed_time(r1) * 1e-3], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        assert torch.isfinite(eng._traj["obs"]).all(), "non-finite observation in the rollout"
        roll = (kk, reps, float(rt[0]), float(rt[1]))

    line = None
    if rank == 0:
        total_lanes = shard.global_lanes
        value = total_lanes * timed_steps / wall_max
        per_launch_s = ev_max / timed_steps  # HIP events on the launch stream, per pf_env_step launch
        out = {
            "metric": (f"env-steps/sec (whole node), QuadX-Hover batch={args.batch} " + ("per GPU" if args.scaling == "weak" else "in total"))
            if args.env == "hover" else f"env-steps/sec (whole node), {args.env}",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "world_size": world, "launcher": ("bench.py started the ranks itself (torch.distributed.run)" if os.environ.get("PF_BENCH_SELF_LAUNCHED") == "1" else
                                              ("torch.distributed.run environment" if "WORLD_SIZE" in os.environ else "single process")),
            "collective_backend": (None if dist is None else ("rccl (torch 'nccl')" if backend == "nccl" else backend)),
            "ms_per_step": 1e3 * wall_max / timed_steps, "higher_is_better": True, "scaling": args.scaling,
            # the timed region: `repeats` x K steps back to back (--min-timed-ms; every figure is per step)
            "timed": {"steps": timed_steps, "repeats": repeats, "wall_ms": 1e3 * wall_max, "event_ms": 1e3 * ev_max, "min_timed_ms": args.min_timed_ms,
                      "per_rank_ms_per_step": None if per_rank is None else [[1e3 * w / timed_steps, 1e3 * e / timed_steps] for w, e in per_rank]},
            # the same quantity from the HIP events around the K launches (excludes the host's graph-launch latency,
            # which a short --steps run amortises over few steps)
            "value_event_timed": total_lanes * timed_steps / ev_max, "nonfinite_lanes": nonfinite,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PyFlyt/QuadX-Hover-v4 semantics, flight_mode 0, batch {n}/GPU x {world} GPU(s), "
                                   f"random actions, motor noise {args.noise}, NEXT_STEP auto-reset"
                       if args.env == "hover" else f"{args.env}, batch {n}/GPU x {world}" + (f", {args.dogfight_actions} actions" if args.env == "dogfight" else ""),
                       "batch_per_gpu": n, "global_batch": total_lanes, "setup": {"untimed_steps_before_warmup": setup_steps, "what": "episode-phase preroll (--preroll), one eager step (kernel load), one replay of the instantiated graph (its upload), clock spin-up rollout right before the warm-up"}, "action_ring": R, "ticks_per_env_step": eng.ticks_per_step,
                       "flight_mode": args.flight_mode, "launch": "hipGraph" if graph is not None else ("prepared steps, one launch per step" if launchers is not None else "eager"), "contact_response": bool(eng.params.contact_response), "world_overrides": args.world, "parallelism": f"dp{world} (independent lanes, no collective)"},
            "roofline": roofline_block(args.env, n, per_launch_s, KERNEL_OF.get(args.env, "pf::quadx_m0_env_kernel")),
        }
        if roll is not None:
            kk, reps, rwall, rev = roll
            per_step = rev / (reps * kk)
            # HBM bytes the rollout really moves per env step and lane: obs + action + reward + 2 flag bytes written,
            # the 7 + 7 state groups once per launch
            obs_b = 4 * eng.obs_dim
            moved = obs_b + 16 + 4 + 2 + (ALGO_BYTES[args.env] - obs_b - 16 - 4 - 2) / kk
            out["rollout"] = {
                "k": kk, "launches": reps, "ms_per_step": 1e3 * per_step, "value": total_lanes / (rwall / (reps * kk)),
                "value_event_timed": total_lanes / per_step,
                # the rollout's HBM fraction: the bytes this launch shape actually moves per env step
                "bytes_per_step_moved": moved, "frac": moved * n / per_step / 1e9 / HBM_PEAK_GBS,
                # NOT a roofline fraction: the one-launch-per-step algorithmic bytes (state round trip included)
                # divided by the rollout's time -- a speed ratio against the per-step launch shape, kept for continuity with r02
                "nominal_vs_per_step_bytes": ALGO_BYTES[args.env] * n / per_step / 1e9 / HBM_PEAK_GBS,
                "kernel": ("pf::quadx_m0_env_kernel" if args.env != "fixedwing_waypoints" else "pf::fixedwing_wp_env_kernel") + "<..., ROLL=1>", "launch_us": rev / reps * 1e6,
                "note": "k env steps per launch, state in registers, on-device action sampling (pf_sample_actions keys); bit-identical to k x pf_env_step (tests/test_gpu_rollout.py)",
            }
        # `value` is the one-launch-per-step figure in EVERY mode (a policy in the loop can reach it): what the driver's scaling curve is
        # computed from. Strong scaling cuts ONE batch over the GPUs -- 65 536 lanes on 8 GPUs are 8 192 per GPU = 128 waves on 1 024
        # SIMDs, a launch per step costs its fixed ~8 us whatever the kernel does (~1.25 x at 8 GPUs) -- and what
        # scales in that regime is the state-resident path: reported under `rollout`, named in `state_resident_note`, never as `value`.
        out["value_kind"] = "one pf_env_step launch per env step (hipGraph replay)"
        if args.scaling == "strong" and roll is not None:
            out["state_resident_note"] = ("strong scaling leaves batch / N lanes per GPU on the launch floor: `rollout` (pf_rollout, k env steps per launch, "
                                          "open-loop or on-device actions) is the figure that scales there; `value` stays the per-step launch")
        if args.env == "hover" and world == 1 and not args.no_configs and args.scaling == "weak" and args.batch == 65536 and args.flight_mode == 0 \
                and not args.no_contact_response and not args.world:
            # BASELINE.json configs 2-4, timed after the headline in the same process (the headline config is configs[4]'s per-GPU
            # slice = 65 536 lanes of Hover; config 1 is the CPU plumbing case: cpu_baseline.single_env_1core)
            out["configs"] = {
                # (one GPU's best roofline point: eight waves per SIMD's worth of lanes, two resident -- 2 000 steps are 0.1 s)
                "hover_524288": time_config_guarded("hover", 524288, device, args),
                "hover_4096": time_config_guarded("hover", 4096, device, args),
                "quadx_waypoints_65536": time_config_guarded("quadx_waypoints", 65536, device, args),
                "fixedwing_waypoints_65536": time_config_guarded("fixedwing_waypoints", 65536, device, args),
            }
            if not args.no_facade:
                try:
                    out["facade"] = facade_block(device, args, ms_per_step=out["ms_per_step"])
                except Exception as e:  # noqa: BLE001  (a reported extra must not take the measured line down)
                    out["facade"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1 and args.env not in ("dogfight", "ma_hover"):
            try:
                out["cpu_baseline"] = cpu_baseline(args.env, args.noise, args.cpu_seconds)
            except Exception as e:  # noqa: BLE001  (the reported baseline must not take the measured line down)
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        line = json.dumps(out)
    if dist is not None:
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes its version banner to the C stdout; flush that first so that the JSON line is the LAST line of output
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
