R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
for cr in 0 1; do CR=$cr TASK=waypoints PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/phase_trace.py 2>&1 | grep -v amdgpu.ids > $O/phase_wp_cr$cr.txt; cat $O/phase_wp_cr$cr.txt; done
