#!/bin/bash
# usage: pmc.sh <tag> [ENV=VAL ...]  -- collects SQ counters for the env kernel, prints per-wave averages
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; shift
env "$@" rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d /tmp/pmc_$tag -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv,collections,glob,sys
tag=sys.argv[1]
for f in glob.glob(f'/tmp/pmc_{tag}/*/*counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'env_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    w=sum(agg['SQ_WAVES'][10:])/len(agg['SQ_WAVES'][10:])
    out={k: sum(v[10:])/len(v[10:])/w for k,v in agg.items()}
    print(tag, 'waves',int(w), ' '.join(f"{k[3:]}={v:.0f}" for k,v in sorted(out.items()) if k!='SQ_WAVES'))
PY
