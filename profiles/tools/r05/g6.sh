#!/bin/bash
# round 5, call 6: Philox4x32-7 (product) against the r04 library, and the machine scheduler ON (variant: 10 rounds, misched) on all three configs
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/build/variants
run() {  # name lib env
  PF_LIB_PATH=$2 timeout 120 python bench.py --env $3 --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $3', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"
}
for rep in 1 2; do
for e in hover quadx_waypoints fixedwing_waypoints; do
  run product_philox7 $GRAFT_REPO_ROOT/pyflyt_amd/libpyflyt_amd.so $e
  run misched_on $V/libpf_misched.so $e
done
done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
