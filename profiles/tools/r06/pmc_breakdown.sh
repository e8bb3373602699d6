#!/bin/bash
# Where a wave's issue slots go, by instruction kind (per config: two SQ passes, no tracing flag) -> gpurun_out/r06_breakdown/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_breakdown; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 170"
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH"
B="SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SENDMSG"
C="SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VSKIPPED"
for e in quadx:hover quadx:waypoints fixedwing:waypoints; do
  v=${e%%:*}; t=${e##*:}
  VEH=$v TASK=$t $T rocprofv3 --pmc $A --output-format csv -d $O/${v}_${t}_a -- python $R/profiles/tools/prof_cfg.py > $O/${v}_${t}_a.log 2>&1
  VEH=$v TASK=$t $T rocprofv3 --pmc $B --output-format csv -d $O/${v}_${t}_b -- python $R/profiles/tools/prof_cfg.py > $O/${v}_${t}_b.log 2>&1
  VEH=$v TASK=$t $T rocprofv3 --pmc $C --output-format csv -d $O/${v}_${t}_c -- python $R/profiles/tools/prof_cfg.py > $O/${v}_${t}_c.log 2>&1
done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
ls $O; tail -3 $O/*_c.log | head -20
