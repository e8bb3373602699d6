#!/bin/bash
# round 5, call 2: Fixedwing-Waypoints on the one-wave-per-SIMD instantiation with the constant table in vector registers and both
# surface pairs evaluated side by side (product build) against round 4's kernel (libpf_fw_w1nocalm.so = r04's tick at 512 registers)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/build/variants
run() {  # name lib env
  PF_LIB_PATH=$2 timeout 120 python bench.py --env $3 --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $3', 'launch_us', round(d['roofline']['launch_us'],2), 'ms_per_step_us', round(d['ms_per_step']*1e3,2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"
}
for rep in 1 2; do
  run product $GRAFT_REPO_ROOT/pyflyt_amd/libpyflyt_amd.so fixedwing_waypoints
  run w1nocalm $V/libpf_fw_w1nocalm.so fixedwing_waypoints
done
timeout 600 python -m pytest tests -x -q -m gpu -k "fixedwing or Fixedwing or fw" 2>&1 | tail -5
