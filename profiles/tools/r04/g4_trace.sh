R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
for t in waypoints; do WHAT=rates TASK=$t RINGS=100 PF_LIB_PATH=$T timeout 200 python $R/profiles/tools/solver_trace.py 2>&1 | grep -v amdgpu.ids; done > $O/rates_quadx.txt
cat $O/rates_quadx.txt
