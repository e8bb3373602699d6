#!/bin/bash
# round 5, call 7: the whole GPU suite on the product (Philox4x32-7, fp64 cascade in the flight modes 1-7), then the cascaded modes' timings
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
for m in 7 6 4; do timeout 100 python bench.py --flight-mode=$m --steps 1000 --warmup 100 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $m', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"; done
