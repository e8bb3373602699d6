#!/bin/bash
# round-3 A/B driver: for each library variant given as NAME=PATH (or NAME= for the in-tree build): headline bench, landed-bodies
# solver bench, dogfight population sweep. Output under gpurun_out/$OUT/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r03ab}; mkdir -p $O
for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  if [ -n "$path" ]; then export PF_LIB_PATH=$R/$path; else unset PF_LIB_PATH; fi
  timeout 120 python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$name.json
  if [ -z "$SKIP_SOLVER" ]; then timeout 120 python $R/profiles/tools/solver_bench.py > $O/solver_$name.txt 2>&1; fi
  if [ -z "$SKIP_DOG" ]; then timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" > $O/dog_$name.txt; fi
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; ro=d.get("rollout") or {}
        print(os.path.basename(f), "launch_us %.2f frac %.3f"%(r["launch_us"], r["frac"]), "rollout us/step %.2f"%(ro.get("ms_per_step",0)*1e3), "CR", d["config"].get("contact_response"))
    except Exception as e: print(f, "ERR", e)
PY
tail -n 3 $O/solver_*.txt 2>/dev/null; tail -n 15 $O/dog_*.txt 2>/dev/null
