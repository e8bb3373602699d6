#!/bin/bash
# A/B on one box: product vs -DPF_KERNARG_WARM (the cold lines of the kernel-argument block requested behind the state loads)
R=$GRAFT_REPO_ROOT; cd $R
one() { python bench.py --env $2 --steps 3000 --warmup 300 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 %.3f us, rollout %.3f us'%(d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))"; }
for i in 1 2 3; do
  one base hover
  PF_LIB_PATH=$R/build/variants/libpf_kw.so one kw hover
done
for i in 1 2; do
one base quadx_waypoints
PF_LIB_PATH=$R/build/variants/libpf_kw.so one kw quadx_waypoints
done
