#!/bin/bash
# the cascaded flight modes' per-step / rollout time on the product library (or PF_LIB_PATH): modes 7, 6, 4, 1, -1 -> gpurun_out/g_modes.txt
cd "$(dirname "$0")/../../.."
out=gpurun_out/g_modes.txt; : > $out
for m in 7 6 4 1 -1; do
  python bench.py --flight-mode $m --steps 1000 --warmup 100 --no-cpu-baseline --no-configs --no-facade 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $m: %.2f us per step (rollout %.2f)' % (d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))" >> $out 2>&1
done
cat $out
