"""Copies the judged artefacts from gpurun_out/r06 into profiles/r06, writes profiles/r06/pmc_summary.json and
profiles/pmc_latest.json (what bench.py replays as roofline.traffic / roofline.issue, keyed by the source hash of the kernels the
counters were collected on) and prints the summary numbers."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
src, dst = os.path.join(R, "gpurun_out", "r06"), os.path.join(R, "profiles", "r06")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "*.txt")):
    shutil.copy(f, dst)
for env in ("hover", "quadx_waypoints", "fixedwing_waypoints"):
    # (gpurun merges a call's files into gpurun_out/: an earlier collection's files, other PIDs in their names, may sit next to them)
    for f in sorted(glob.glob(os.path.join(src, "kt_" + env, "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        shutil.copy(f, os.path.join(dst, f"rocprofv3_kernel_stats_bench_{env}65536.csv"))


for f in sorted(glob.glob(os.path.join(src, "kt_facade", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
    shutil.copy(f, os.path.join(dst, "rocprofv3_kernel_stats_facade_closed_loop.csv"))


def counters(tag, skip, match="env_kernel"):
    out, meta = {}, {}
    for f in sorted(glob.glob(os.path.join(src, tag, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:  # (the newest run only)
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64 * 64:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
        for k, v in agg.items():
            v = v[skip:]
            out[k] = {"avg_per_launch": sum(v) / max(1, len(v)), "launches": len(v)}
    return out, meta


ALGO = {"hover": 330, "quadx_waypoints": 442, "fixedwing_waypoints": 418}
summary, latest = {}, {}
for env, (v, t) in {"hover": ("quadx", "hover"), "quadx_waypoints": ("quadx", "waypoints"), "fixedwing_waypoints": ("fixedwing", "waypoints")}.items():
    fetch, meta = counters(f"pmc_{v}_{t}_FETCH_SIZE", 10)
    write, _ = counters(f"pmc_{v}_{t}_WRITE_SIZE", 10)
    sq, _ = counters(f"pmc_{v}_{t}_sq", 10)
    if not fetch or not write or not sq:
        print("missing counters for", env)
        continue
    fe, wr = fetch["FETCH_SIZE"]["avg_per_launch"], write["WRITE_SIZE"]["avg_per_launch"]
    waves = sq["SQ_WAVES"]["avg_per_launch"]
    per_wave = {k[3:]: x["avg_per_launch"] / waves for k, x in sq.items() if k != "SQ_WAVES"}
    summary[env] = {
        "kernel": meta, "note": "rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
                                "(MI355X_MICROARCH.md, HBM section) -> doubled; separate --pmc passes, no tracing flags",
        "read_bytes_per_launch": 2 * fe * 1024, "write_bytes_per_launch": wr * 1024, "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
        "algorithmic_bytes_per_launch": ALGO[env] * 65536, "per_wave_per_env_step": per_wave, "launches_averaged": fetch["FETCH_SIZE"]["launches"],
    }
    insts = per_wave.get("INSTS_VALU", 0.0) + per_wave.get("INSTS_SALU", 0.0)
    latest[env] = {"batch": 65536, "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "valu_per_wave": per_wave.get("INSTS_VALU"),
                   "salu_per_wave": per_wave.get("INSTS_SALU"),
                   # (quad-cycles in which the wave issued anything: VALU + SALU + LDS + VMEM + SMEM + branches / waits, a transcendental
                   #  counted twice -- what a LONE wave pays four clocks each for, profiles/r06/lone_wave_issue.txt)
                   "issue_slots_per_wave": per_wave.get("ACTIVE_INST_ANY"),
                   # (SQ_WAVE_CYCLES under counter collection is not the product's wave life: 46 k clocks per Hover wave in this
                   #  collection against 16 k by the phase trace -- the counters' own run time; not replayed)
                   "clocks_per_inst": None}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
import bench  # noqa: E402

json.dump({"source_hash": bench.source_hash(), "source": "profiles/r06/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_*, separate passes, FETCH doubled per guide)",
           "envs": latest}, open(os.path.join(R, "profiles", "pmc_latest.json"), "w"), indent=1)
for f in sorted(glob.glob(os.path.join(dst, "bench_*.json"))):
    try:
        d = json.load(open(f))
    except Exception:
        print(os.path.basename(f), "unreadable"); continue
    r = d["roofline"]; ro = d.get("rollout")
    print(os.path.basename(f), "value %.3e" % d["value"], "launch_us %.2f frac %.3f" % (r["launch_us"], r["frac"]),
          ("| rollout %.2f us/step" % (ro["ms_per_step"] * 1e3)) if ro else "", "| cpu", d.get("cpu_baseline", {}).get("value"))
    for k, c in (d.get("configs") or {}).items():
        print("    config", k, "launch_us %.2f frac %.3f" % (c["launch_us"], c["roofline"]["frac"]))
for env in ("hover", "quadx_waypoints", "fixedwing_waypoints"):
    p = os.path.join(dst, f"rocprofv3_kernel_stats_bench_{env}65536.csv")
    if os.path.exists(p):
        print(env, open(p).read().split("\n")[1][:160])
print(json.dumps(latest, indent=1))
