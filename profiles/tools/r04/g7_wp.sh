R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
WHAT=rates TASK=waypoints RINGS=100 PF_LIB_PATH=$T timeout 200 python $R/profiles/tools/solver_trace.py 2>&1 | grep -v amdgpu.ids > $O/rates_wp.txt; cat $O/rates_wp.txt
TASK=waypoints PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/phase_trace.py 2>&1 | grep -v amdgpu.ids > $O/phase_wp.txt; tail -6 $O/phase_wp.txt
timeout 170 rocprofv3 --kernel-trace --output-format csv -d $O/kt_wp -- python $R/bench.py --env quadx_waypoints --steps 1000 --warmup 100 --no-cpu-baseline --rollout-steps 0 > /dev/null 2>&1
python - <<'PY'
import csv,glob,os,numpy as np
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04g/kt_wp/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'quadx_m0_env_kernel' in r['Kernel_Name']]
d=np.array(d[-1000:])
print("waypoints launches", len(d), "min %.2f p10 %.2f median %.2f mean %.2f p90 %.2f p99 %.2f max %.2f"%(d.min(),np.percentile(d,10),np.median(d),d.mean(),np.percentile(d,90),np.percentile(d,99),d.max()))
print("histogram (us):", np.histogram(d, bins=[0,12,14,16,18,20,24,28,32,40,60,100])[0])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
