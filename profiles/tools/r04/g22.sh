#!/bin/bash
# same-box A/B of the prologue / epilogue latency changes (variant libraries A..E, see profiles/README.md)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in A B C D E; do
  for e in hover quadx_waypoints; do
  PF_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/libpf_$v.so timeout 100 python bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $e', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round(d.get('rollout',{}).get('ms_per_step')*1e3,2))"
  done
done
done
