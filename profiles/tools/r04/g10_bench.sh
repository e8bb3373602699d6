R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(time timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5) > $O/driver.json 2> $O/driver.err; tail -3 $O/driver.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04h/driver.json').read().strip().split("\n")[-1])
print({k:d[k] for k in ("value","ms_per_step","value_event_timed","timed")})
print(d["roofline"])
for k,c in d.get("configs",{}).items(): print(k, round(c["launch_us"],2), round(c["roofline"]["frac"],3))
print(d.get("cpu_baseline",{}).get("value"))
PY
