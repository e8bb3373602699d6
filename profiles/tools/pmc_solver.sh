#!/bin/bash
# SQ counters of the Aviary step with every body resting on the floor (profiles/tools/solver_bench.py): where the contact solve spends its time
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
N=4096 timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/pmc_solver -- python $R/profiles/tools/solver_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,collections,glob
for f in glob.glob('/tmp/pmc_solver/*/*counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'aviary_step_kernel' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items():
        out={n: sum(v[-100:])/len(v[-100:]) for n,v in c.items()}
        w=out.get('SQ_WAVES',1)
        print(k, 'waves', int(w), 'per wave per Aviary step (2 ticks):', ' '.join(f"{n[3:]}={v/w:.0f}" for n,v in sorted(out.items()) if n!='SQ_WAVES'))
PY
