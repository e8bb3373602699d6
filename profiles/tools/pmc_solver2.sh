#!/bin/bash
# SQ counters of the Aviary step with every body resting on the floor (solver_bench.py), two passes (issue / wait, LDS)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_solver_$i
  N=4096 timeout 150 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_solver_$i -- python $R/profiles/tools/solver_bench.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv,collections,glob
for f in sorted(glob.glob('/tmp/pmc_solver_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'aviary_step_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:46]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
        out={n: sum(v[-100:])/len(v[-100:]) for n,v in c.items()}
        w=out.get('SQ_WAVES',1)
        print(k, 'per wave per Aviary step (2 ticks):', ' '.join(f"{n[3:]}={v/w:.0f}" for n,v in sorted(out.items()) if n!='SQ_WAVES'))
PY
