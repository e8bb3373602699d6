#!/bin/bash
# A/B of library variants on the Fixedwing-Waypoints bench: fw_ab.sh NAME=PATH ...
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  if [ -n "$path" ]; then export PF_LIB_PATH=$R/$path; else unset PF_LIB_PATH; fi
  timeout 120 python $R/bench.py --env fixedwing_waypoints --steps 500 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; ro=d.get('rollout') or {}
print('$name', 'launch_us %.2f frac %.3f'%(r['launch_us'], r['frac']), 'rollout us/step %.2f'%(ro.get('ms_per_step',0)*1e3))"
done
