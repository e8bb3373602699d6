#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_onestep.py -q -s -k "mode7 or mode or one_step" 2>&1 | grep -E "mode|one step|passed|failed|FAILED|worst" | tail -40
for m in 7 6 4; do timeout 100 python bench.py --flight-mode=$m --steps 1000 --warmup 100 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $m', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"; done
for e in hover quadx_waypoints fixedwing_waypoints; do timeout 100 python bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"; done
