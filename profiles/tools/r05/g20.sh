#!/bin/bash
# A/B on one box: product (constant table through LDS; the rare code touched by one branch-free load per wave) vs the same without
# any touch of the rare code (nowarm), vs the previous table-through-LDS build with the prefetch simply removed (fwln, Fixedwing only)
R=$GRAFT_REPO_ROOT; cd $R
one() { python bench.py --env $2 --steps 2500 --warmup 250 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 %.3f us, rollout %.3f us'%(d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))"; }
for i in 1 2; do
  for e in fixedwing_waypoints hover quadx_waypoints; do
    one new $e
    PF_LIB_PATH=$R/build/variants/libpf_nowarm.so one nowarm $e
  done
  PF_LIB_PATH=$R/build/variants/libpf_fwln.so one fwln fixedwing_waypoints
done
timeout 600 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_golden.py tests/test_gpu_calm_path.py tests/test_gpu_spares.py -q -x 2>&1 | tail -3
