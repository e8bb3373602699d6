#!/bin/bash
# A/B of Fixedwing-Waypoints variant libraries on ONE box (same clocks, same neighbours): build/variants/fw_*.so, two passes each.
# Usage (through gpurun): bash profiles/tools/r06/g_fw.sh [lib ...]      -> gpurun_out/g_fw.txt
cd "$(dirname "$0")/../../.."
out=gpurun_out/g_fw.txt; : > $out
libs="$@"; [ -z "$libs" ] && libs=$(ls build/variants/fw_*.so)
for pass in 1 2; do
  for lib in $libs; do
    PF_LIB_PATH=$PWD/$lib python bench.py --env fixedwing_waypoints --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --rollout-steps 100 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib pass $pass: %.3f us per step (events %.3f), rollout %.3f us' % (d['ms_per_step']*1e3, d['roofline']['launch_us'], d['rollout']['ms_per_step']*1e3))" >> $out 2>&1
  done
done
cat $out
