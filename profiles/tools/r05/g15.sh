#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for e in quadx_waypoints hover; do
timeout 170 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$e -- python $R/bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --rollout-steps 0 > /dev/null 2>&1
python - <<PY
import csv,glob,os,numpy as np
f=glob.glob('$O/kt_$e/*/*kernel_trace.csv')[0]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'quadx_m0_env_kernel' in r['Kernel_Name'] ]
d=np.array(d[-2000:]); d=d[d>0]
print("$e launches", len(d), "min %.2f p10 %.2f median %.2f mean %.2f p90 %.2f p99 %.2f max %.2f std %.2f"%(d.min(),np.percentile(d,10),np.median(d),d.mean(),np.percentile(d,90),np.percentile(d,99),d.max(),d.std()))
print("histogram (us):", np.histogram(d, bins=[0,9,10,11,12,13,14,15,16,17,18,20,24,28,32,40,60,100])[0].tolist())
PY
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
