#!/bin/bash
# A/B of QuadX-Waypoints variant libraries on ONE box: bench (mean launch) over 6000 steps, two passes each -> gpurun_out/g_qw.txt
cd "$(dirname "$0")/../../.."
out=gpurun_out/g_qw.txt; : > $out
for pass in 1 2; do
  for lib in "$@"; do
    PF_LIB_PATH=$PWD/$lib python bench.py --env quadx_waypoints --steps 6000 --warmup 200 --no-cpu-baseline --no-configs --rollout-steps 100 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib pass $pass: %.3f us per step (events %.3f), rollout %.3f us' % (d['ms_per_step']*1e3, d['roofline']['launch_us'], d['rollout']['ms_per_step']*1e3))" >> $out 2>&1
  done
done
cat $out
