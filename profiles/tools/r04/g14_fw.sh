R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for i in 1 2; do timeout 100 python $R/bench.py --env fixedwing_waypoints --steps 1000 --warmup 100 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fixedwing launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round(d.get('rollout',{}).get('ms_per_step',0)*1e3,2))"; done
cd $R; timeout 600 python -m pytest tests -m gpu -q --tb=short -k "fixedwing or dogfight or rollout or fullsize" 2>&1 | tail -5
