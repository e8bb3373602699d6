#!/bin/bash
# action prefetch A/B: the three 65 536-lane configs + 4 096 + the bit-identity / parity tests that cover the per-step kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h
for e in hover quadx_waypoints fixedwing_waypoints; do
  timeout 100 python bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', 'launch_us', round(d['roofline']['launch_us'],2), 'wall_us', round(d['ms_per_step']*1e3,2), 'rollout', d.get('rollout',{}).get('ms_per_step'))"
done
timeout 100 python bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hover 4096', 'launch_us', round(d['roofline']['launch_us'],2))"
timeout 100 python bench.py --batch 524288 --steps 300 --warmup 50 --no-cpu-baseline --no-configs --rollout-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hover 524288', 'launch_us', round(d['roofline']['launch_us'],2), d['value'])"
timeout 900 python -m pytest tests -q -m gpu -x -k "rollout or parity or golden or calm or api or fullsize" 2>&1 | tail -3
