R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for e in hover quadx_waypoints fixedwing_waypoints; do timeout 100 python $R/bench.py --env $e --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$e.json; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04c/bench_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), "launch_us", round(d['roofline']['launch_us'],2), "wall_us", round(d['ms_per_step']*1e3,2), "rollout_us", round(d.get('rollout',{}).get('ms_per_step',0)*1e3,2))
    except Exception as ex: print(f, ex)
PY
