#!/bin/bash
# Hover / Waypoints with the floor solve's constants behind the device parameter block, and the tests the change touches
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do python bench.py --env hover --steps 3000 --warmup 300 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hover', d['ms_per_step']*1e3, d.get('rollout'))"; done
python bench.py --env quadx_waypoints --steps 3000 --warmup 300 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wp', d['ms_per_step']*1e3, d.get('rollout'))"
timeout 900 python -m pytest tests/test_gpu_calm_path.py tests/test_gpu_rollout.py tests/test_gpu_spares.py tests/test_gpu_parity.py tests/test_gpu_onestep.py -x -q 2>&1 | tail -5
