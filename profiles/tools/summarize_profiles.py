"""Copies the judged artefacts from gpurun_out/r01 into profiles/r01 and prints the summary numbers."""
import csv, collections, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(R, "gpurun_out", "r01"), os.path.join(R, "profiles", "r01")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, dst)
for f in glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "rocprofv3_kernel_stats_bench_hover65536.csv"))
for f in glob.glob(os.path.join(src, "kt", "*", "*domain_stats.csv")):
    shutil.copy(f, os.path.join(dst, "rocprofv3_domain_stats_bench_hover65536.csv"))
for env in ("quadx_waypoints", "fixedwing_waypoints"):
    for f in glob.glob(os.path.join(src, "kt_" + env, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"rocprofv3_kernel_stats_bench_{env}65536.csv"))
out = {}
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq", "pmc_tcc"):
    for f in glob.glob(os.path.join(src, d, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list); meta = {}
        for r in csv.DictReader(open(f)):
            if "env_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
        for k, v in agg.items():
            v = v[10:]; out[k] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
        out["_kernel"] = meta
fetch, write = out["FETCH_SIZE"]["avg_per_launch"], out["WRITE_SIZE"]["avg_per_launch"]
out["_derived"] = {
    "note": "rocprofv3 FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> doubled",
    "read_bytes_per_launch": 2 * fetch * 1024, "write_bytes_per_launch": write * 1024, "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
    "algorithmic_bytes_per_launch": 330 * 65536,
    "per_wave": {k[3:]: out[k]["avg_per_launch"] / out["SQ_WAVES"]["avg_per_launch"] for k in out if k.startswith("SQ_")}}
json.dump(out, open(os.path.join(dst, "pmc_summary_hover65536.json"), "w"), indent=1)
json.dump({"env": "hover", "batch": 65536, "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
           "source": "profiles/r01/pmc_summary_hover65536.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled per guide)"},
          open(os.path.join(R, "profiles", "pmc_latest.json"), "w"), indent=1)
other = {}
for env in ("fixedwing_waypoints", "quadx_waypoints"):
    for f in glob.glob(os.path.join(src, "pmc_sq_" + env, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list); name = ""
        for r in csv.DictReader(open(f)):
            if "env_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"]
        w = sum(agg["SQ_WAVES"][10:]) / len(agg["SQ_WAVES"][10:])
        other[env] = {"kernel": name.split("(")[0], "waves": w,
                      "per_wave": {k[3:]: sum(v[10:]) / len(v[10:]) / w for k, v in agg.items() if k != "SQ_WAVES"}}
if other:
    json.dump(other, open(os.path.join(dst, "pmc_sq_other_kernels65536.json"), "w"), indent=1)
    print(json.dumps(other, indent=1))
for f in sorted(glob.glob(os.path.join(dst, "bench_*.json"))):
    d = json.load(open(f)); r = d["roofline"]
    print(os.path.basename(f), "value %.3e" % d["value"], "launch_us %.2f" % r["launch_us"], "achieved %.0f GB/s frac %.3f" % (r["achieved"], r["frac"]),
          "cpu", d.get("cpu_baseline", {}).get("value"))
print(open(os.path.join(dst, "rocprofv3_kernel_stats_bench_hover65536.csv")).read().split("\n")[1][:200])
print(json.dumps(out["_derived"], indent=1))
