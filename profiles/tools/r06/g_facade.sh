#!/bin/bash
# A/B of full library variants on the facade block (make_vec(...).step(), MAQuadXHoverEnv.step()) on ONE box, two passes -> gpurun_out/g_facade.txt
cd "$(dirname "$0")/../../.."
out=gpurun_out/g_facade.txt; : > $out
for pass in 1 2; do
  for lib in "$@"; do
    PF_LIB_PATH=$PWD/$lib python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['facade']; print('$lib pass $pass: launch %.3f us | VectorEnv.step eager %.2f graph %.2f | PettingZoo step eager %.2f graph %.2f' % (d['ms_per_step']*1e3, f['vector_env']['step_only']['eager_us_per_step'], f['vector_env']['step_only']['graph_us_per_step'], f['pettingzoo']['step_only']['eager_us_per_step'], f['pettingzoo']['step_only']['graph_us_per_step']))" >> $out 2>&1
  done
done
cat $out
