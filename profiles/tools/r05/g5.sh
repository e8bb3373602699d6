#!/bin/bash
# round 5, call 5: Fixedwing-Waypoints after the load reorder; instruction counters of the new kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
timeout 120 python $R/bench.py --env fixedwing_waypoints --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('product fixedwing', 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round((d.get('rollout') or {}).get('ms_per_step',0)*1e3,2))"
done
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
VEH=fixedwing TASK=waypoints timeout 170 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_fw_sq -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
python - <<'PY'
import csv,glob,os,collections
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05/pmc_fw_sq/*/*counter_collection.csv')
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'fixedwing_wp_env_kernel' in r['Kernel_Name'] and int(r['Grid_Size'])>=65536:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
for k,v in acc.items(): print(k, round(sum(v)/len(v)/w,1), 'per wave' if k!='SQ_WAVES' else '', len(v))
PY
find $O -name "*.db" -delete
