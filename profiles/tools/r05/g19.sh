#!/bin/bash
# A/B on one box, Fixedwing-Waypoints: product vs the constant table through LDS (two word loads per lane + 28 broadcast ds_read_b128
# instead of 28 broadcast global loads), with and without the rare-code prefetch of the first 16 workgroups
R=$GRAFT_REPO_ROOT; cd $R
one() { python bench.py --env fixedwing_waypoints --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.3f us, rollout %.3f us'%(d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))"; }
for i in 1 2; do
  one base
  PF_LIB_PATH=$R/build/variants/libpf_fwl.so one via_lds
  PF_LIB_PATH=$R/build/variants/libpf_fwln.so one via_lds_no_code_warm
  PF_LIB_PATH=$R/build/variants/libpf_fwn.so one no_code_warm
done
PF_LIB_PATH=$R/build/variants/libpf_fwln.so timeout 300 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_golden.py -q -x -k "fixedwing or fw" 2>&1 | tail -3
