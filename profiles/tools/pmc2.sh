#!/bin/bash
# usage: pmc2.sh <tag> "<counters>" [ENV=VAL ...] -- arbitrary SQ counters for the env kernel, per-wave averages
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; ctrs=$2; shift; shift
env "$@" timeout 150 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv,collections,glob,sys
tag=sys.argv[1]
for f in glob.glob(f'/tmp/pmc_{tag}/*/*counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'env_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(tag, 'per wave:', ' '.join(f"{k[3:]}={sum(v[10:])/len(v[10:])/1024:.0f}" for k,v in sorted(agg.items())))
PY
