#!/bin/bash
# round 5, call 4: per-wave phase timelines (PF_PHASE_TRACE variant library; the Fixedwing per-tick atomics have their own switch now)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T=$R/build/variants/libpf_trace.so
VEH=fixedwing TASK=waypoints PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/phase_trace.py 2>&1 | grep -v amdgpu.ids > $O/phase_fixedwing.txt
for t in hover waypoints; do TASK=$t PF_LIB_PATH=$T timeout 100 python $R/profiles/tools/phase_trace.py 2>&1 | grep -v amdgpu.ids > $O/phase_$t.txt; done
cat $O/phase_fixedwing.txt $O/phase_hover.txt $O/phase_waypoints.txt
