#!/bin/bash
# A/B on one box, Fixedwing-Waypoints: the table's LDS stage behind the Philox call (base) vs in front of it, its words requested
# first and the seeds' scalar load waited for in front of the LDS reads (fw2)
R=$GRAFT_REPO_ROOT; cd $R
one() { python bench.py --env fixedwing_waypoints --steps 2500 --warmup 250 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 %.3f us, rollout %.3f us'%(d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))"; }
for i in 1 2 3; do
  one base
  PF_LIB_PATH=$R/build/variants/libpf_fw2.so one fw2
done
PF_LIB_PATH=$R/build/variants/libpf_fw2.so timeout 300 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_golden.py tests/test_gpu_parity.py -q -x -k "fixedwing or fw" 2>&1 | tail -2
