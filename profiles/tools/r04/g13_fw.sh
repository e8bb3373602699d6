R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
VEH=fixedwing TASK=waypoints PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/phase_trace.py 2>&1 | grep -v amdgpu.ids > $O/phase_fw.txt; cat $O/phase_fw.txt
