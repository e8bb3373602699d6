"""Copies the judged artefacts from gpurun_out/r03 into profiles/r03 and prints / stores the summary numbers."""
import collections
import csv
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(R, "gpurun_out", "r03"), os.path.join(R, "profiles", "r03")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, dst)
for tag, name in (("kt", "hover65536"), ("kt_quadx_waypoints", "quadx_waypoints65536"), ("kt_fixedwing_waypoints", "fixedwing_waypoints65536"),
                  ("kt_dogfight", "dogfight65536"), ("kt_ma_hover", "ma_hover_shared65536"), ("kt_mode7", "hover_mode7_65536")):
    for f in glob.glob(os.path.join(src, tag, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"rocprofv3_kernel_stats_bench_{name}.csv"))
    for f in glob.glob(os.path.join(src, tag, "*", "*domain_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"rocprofv3_domain_stats_bench_{name}.csv"))


def counters(tag, skip, match="env_kernel"):
    out, meta = {}, {}
    for f in glob.glob(os.path.join(src, tag, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64 * 64:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
        for k, v in agg.items():
            v = v[skip:]
            out[k] = {"avg_per_launch": sum(v) / max(1, len(v)), "launches": len(v)}
    return out, meta


summary = {}
# per-step launches (prof_cfg.py: 60 eager env steps; the first 10 skipped) and rollout launches (prof_roll.py: reset + 6 launches
# of 100 steps; the reset launch and the first rollout skipped)
for mode, skip, steps_per_launch in (("step", 10, 1), ("roll", 2, 100)):
    fetch, meta = counters(f"pmc_{mode}_FETCH_SIZE", skip)
    write, _ = counters(f"pmc_{mode}_WRITE_SIZE", skip)
    sq, _ = counters(f"pmc_{mode}_sq", skip)
    fe, wr = fetch["FETCH_SIZE"]["avg_per_launch"], write["WRITE_SIZE"]["avg_per_launch"]
    waves = sq["SQ_WAVES"]["avg_per_launch"]
    summary[mode] = {
        "kernel": meta, "env_steps_per_launch": steps_per_launch,
        "note": "rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> doubled",
        "read_bytes_per_env_step": 2 * fe * 1024 / steps_per_launch, "write_bytes_per_env_step": wr * 1024 / steps_per_launch,
        "hbm_bytes_per_env_step": (2 * fe + wr) * 1024 / steps_per_launch, "algorithmic_bytes_per_env_step": 330 * 65536,
        "per_wave_per_env_step": {k[3:]: v["avg_per_launch"] / waves / steps_per_launch for k, v in sq.items() if k != "SQ_WAVES"},
        "launches_averaged": fetch["FETCH_SIZE"]["launches"],
    }
for tag in ("pmc_step_sq_detect_only", "pmc_step_sq_mode7", "pmc_step_sq_fixedwing_waypoints", "pmc_step_sq_quadx_waypoints"):
    sq, meta = counters(tag, 10)
    if sq:
        waves = sq["SQ_WAVES"]["avg_per_launch"]
        summary[tag[len("pmc_step_sq_"):]] = {"kernel": meta.get("Kernel_Name", "").split("(")[0], "VGPR": meta.get("VGPR_Count"), "waves": waves,
                                             "per_wave_per_env_step": {k[3:]: v["avg_per_launch"] / waves for k, v in sq.items() if k != "SQ_WAVES"}}
json.dump(summary, open(os.path.join(dst, "pmc_summary_hover65536.json"), "w"), indent=1)
json.dump({"env": "hover", "batch": 65536, "hbm_bytes_per_launch": summary["step"]["hbm_bytes_per_env_step"],
           "source": "profiles/r03/pmc_summary_hover65536.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per guide)"},
          open(os.path.join(R, "profiles", "pmc_latest.json"), "w"), indent=1)
for f in ("dogfight_step_time_vs_population.txt", "solver_bench_landed.txt", "ma_hover_shared_step_time.txt", "phase_trace_hover65536_cr1.txt",
          "phase_trace_hover65536_cr0.txt", "solver_trace.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), dst)
for f in sorted(glob.glob(os.path.join(dst, "bench_*.json"))):
    d = json.load(open(f)); r = d["roofline"]; ro = d.get("rollout")
    print(os.path.basename(f), "value %.3e" % d["value"], "launch_us %.2f frac %.3f" % (r["launch_us"], r["frac"]),
          ("| rollout %.2f us/step frac(moved) %.3f" % (ro["ms_per_step"] * 1e3, ro["frac"])) if ro else "", "| cpu", d.get("cpu_baseline", {}).get("value"))
print(open(os.path.join(dst, "rocprofv3_kernel_stats_bench_hover65536.csv")).read().split("\n")[1][:120])
print(open(os.path.join(dst, "rocprofv3_kernel_stats_bench_hover65536.csv")).read().split("\n")[2][:120])
print(json.dumps({k: (v.get("hbm_bytes_per_env_step"), v["per_wave_per_env_step"]) for k, v in summary.items()}, indent=1))
