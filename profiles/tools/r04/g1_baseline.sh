R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
for t in waypoints hover; do WHAT=rates TASK=$t RINGS=0,100 PF_LIB_PATH=$T timeout 200 python $R/profiles/tools/solver_trace.py 2>&1 | grep -v amdgpu.ids; done > $O/rates_quadx.txt
WHAT=rates VEH=fixedwing TASK=waypoints RINGS=100 PF_LIB_PATH=$T timeout 200 python $R/profiles/tools/solver_trace.py 2>&1 | grep -v amdgpu.ids > $O/rates_fw.txt
WHAT=calm TASK=waypoints PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/solver_trace.py 2>&1 | grep -v amdgpu.ids > $O/calm_wp.txt
for e in hover quadx_waypoints fixedwing_waypoints; do timeout 100 python $R/bench.py --env $e --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$e.json; done
cat $O/rates_quadx.txt $O/rates_fw.txt $O/calm_wp.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04a/bench_*.json')):
    d=json.load(open(f)); print(os.path.basename(f), d['roofline']['launch_us'], d['ms_per_step']*1e3, d.get('rollout',{}).get('ms_per_step'))
PY
