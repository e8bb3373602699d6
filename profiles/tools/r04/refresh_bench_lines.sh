#!/bin/bash
# the two driver-shaped bench lines again, AFTER summarize_profiles_r04.py has rewritten profiles/pmc_latest.json for the final kernels
# (inside the collection they still see the previous file and report its traffic as stale)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 2000 --warmup 200 2>/dev/null | tail -1 > $O/bench_n1.json
timeout 200 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_shape.json
python -c "
import json
for f in ('bench_n1','bench_driver_shape'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['frac'], r['traffic'], r.get('traffic_source'), r.get('issue'))
    for k,v in d.get('configs',{}).items(): print('  ',k,v['launch_us'],v['roofline']['frac'],v['roofline'].get('traffic'))
"
