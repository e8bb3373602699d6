#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for e in hover quadx_waypoints; do timeout 100 python bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"; done; done
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -12
