R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
WHAT=perlaunch TASK=waypoints LAUNCHES=1500 PF_LIB_PATH=$T timeout 300 python $R/profiles/tools/solver_trace.py 2>&1 | grep -v amdgpu.ids > $O/perlaunch_wp.txt; cat $O/perlaunch_wp.txt
TASK=waypoints PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/phase_trace.py 2>&1 | grep -v amdgpu.ids > $O/phase_wp.txt; tail -6 $O/phase_wp.txt
