#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/gputest_plain.log
bash profiles/tools/r06/g_all.sh build/variants/r06_head.so pyflyt_amd/libpyflyt_amd.so > /dev/null 2>&1
for lib in build/variants/r06_head.so pyflyt_amd/libpyflyt_amd.so; do for pass in 1 2; do
  PF_LIB_PATH=$PWD/$lib python bench.py --env quadx_waypoints --batch 524288 --steps 1000 --warmup 100 --no-cpu-baseline --no-configs --no-facade 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib pass $pass waypoints 524288: %.2f us (rollout %.2f)' % (d['ms_per_step']*1e3, d['rollout']['ms_per_step']*1e3))" >> gpurun_out/g_all.txt 2>&1
done; done
MODE=7 TASK=hover PF_LIB_PATH=$PWD/build/variants/t_hover_modes.so timeout 200 python profiles/tools/phase_trace.py > gpurun_out/phase_trace_mode7.txt 2>&1
MODE=0 TASK=hover PF_LIB_PATH=$PWD/build/variants/t_hover_modes.so timeout 200 python profiles/tools/phase_trace.py > gpurun_out/phase_trace_modes_mode0.txt 2>&1
cat gpurun_out/gputest_plain.log gpurun_out/g_all.txt gpurun_out/phase_trace_mode7.txt
