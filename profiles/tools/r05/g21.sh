#!/bin/bash
# A/B on one box: product (observation tile: every LDS read ahead of the first store; Fixedwing: table through LDS, no rare-code
# prefetch at one wave per SIMD) vs the same with the old tile loop (notile) vs without the QuadX kernels' rare-code prefetch (nowarm)
R=$GRAFT_REPO_ROOT; cd $R
one() { python bench.py --env $2 --steps 2500 --warmup 250 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 %.3f us, rollout %.3f us'%(d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))"; }
for i in 1 2; do
  for e in fixedwing_waypoints hover quadx_waypoints; do
    one new $e
    PF_LIB_PATH=$R/build/variants/libpf_notile.so one notile $e
    PF_LIB_PATH=$R/build/variants/libpf_nowarm.so one nowarm $e
  done
done
timeout 600 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_golden.py tests/test_gpu_calm_path.py tests/test_gpu_spares.py tests/test_gpu_api.py -q -x 2>&1 | tail -3
