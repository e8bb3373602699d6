#!/bin/bash
# A/B on one box: HEAD library vs the floor solve's constants out of the kernel arguments (behind the device parameter block: "new";
# derived in the kernel from the block: "v1"), interleaved runs
R=$GRAFT_REPO_ROOT; cd $R
one() { python bench.py --env $2 --steps 3000 --warmup 300 --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 %.3f us, rollout %.3f us'%(d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))"; }
for i in 1 2 3; do
  PF_LIB_PATH=$R/build/variants/libpf_head.so one head hover
  PF_LIB_PATH=$R/build/variants/libpf_v1.so one v1 hover
  one new hover
done
for i in 1 2; do
PF_LIB_PATH=$R/build/variants/libpf_head.so one head quadx_waypoints
PF_LIB_PATH=$R/build/variants/libpf_v1.so one v1 quadx_waypoints
one new quadx_waypoints
done
