#!/bin/bash
# per-launch SQ counters of the dogfight kernel over profiles/tools/dog_diag.py's 700 steps (FREEZE=1 for freeze_wrecks), averaged per 50 launches
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmc_dog -- python $R/profiles/tools/dog_diag.py > /dev/null 2>&1
python3 - <<'PY'
import csv,collections,glob
for f in glob.glob('/tmp/pmc_dog/*/*counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'dogfight_env_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    n=len(agg['SQ_WAVES'])
    for b in range(0, n, 50):
        w=sum(agg['SQ_WAVES'][b:b+50])
        print(f"launches {b:4d}-{b+49:4d}: " + ' '.join(f"{k[3:]}={sum(v[b:b+50])/w:.0f}" for k,v in sorted(agg.items()) if k!='SQ_WAVES'))
PY
