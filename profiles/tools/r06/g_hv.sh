#!/bin/bash
# A/B of QuadX-Hover variant libraries on ONE box at 65 536 and 4 096 lanes, two passes each -> gpurun_out/g_hv.txt
cd "$(dirname "$0")/../../.."
out=gpurun_out/g_hv.txt; : > $out
for pass in 1 2; do
  for lib in "$@"; do
    for b in 65536 4096; do
    PF_LIB_PATH=$PWD/$lib python bench.py --batch $b --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --rollout-steps 100 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib pass $pass batch $b: %.3f us per step (events %.3f), rollout %.3f us' % (d['ms_per_step']*1e3, d['roofline']['launch_us'], d['rollout']['ms_per_step']*1e3))" >> $out 2>&1
    done
  done
done
cat $out
