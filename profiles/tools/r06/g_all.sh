#!/bin/bash
# Same-box A/B of two FULL libraries over the four BASELINE configurations (bench.py's secondary configs): two passes each.
# Usage (through gpurun): bash profiles/tools/r06/g_all.sh libA.so libB.so -> gpurun_out/g_all.txt
cd "$(dirname "$0")/../../.."
out=gpurun_out/g_all.txt; : > $out
for pass in 1 2; do
  for lib in "$@"; do
    PF_LIB_PATH=$PWD/$lib python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-facade 2>/dev/null | tail -1 | \
      python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['configs']
print('$lib pass $pass: hover %.3f (rollout %.3f) | 524288: %.2f | 4096: %.3f (rollout %.3f) | waypoints %.3f (rollout %.3f) | fixedwing %.3f (rollout %.3f)' % (
  d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3, c['hover_524288']['launch_us'], c['hover_4096']['launch_us'], c['hover_4096']['rollout']['us_per_step'],
  c['quadx_waypoints_65536']['launch_us'], c['quadx_waypoints_65536']['rollout']['us_per_step'], c['fixedwing_waypoints_65536']['launch_us'], c['fixedwing_waypoints_65536']['rollout']['us_per_step']))" >> $out 2>&1
  done
done
cat $out
