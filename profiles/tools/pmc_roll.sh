#!/bin/bash
# usage: pmc_roll.sh <tag> "<counters>" [ENV=VAL ...] -- SQ counters of the pf_rollout launches, per wave and per env step
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; ctrs=$2; shift; shift
env "$@" timeout 150 rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$tag -- python $R/profiles/tools/prof_roll.py > /dev/null 2>&1
K=100; for kv in "$@"; do case $kv in K=*) K=${kv#K=};; esac; done
python3 - "$tag" "$K" <<'PY'
import csv,collections,glob,sys
tag,K=sys.argv[1],int(sys.argv[2])
for f in glob.glob(f'/tmp/pmc_{tag}/*/*counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'env_kernel' in r['Kernel_Name'] and int(r['Grid_Size'])>=64*64: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    # the first matching launch is env_reset (one step), the rest are rollouts: keep launches 2..
    out={k: sum(v[2:])/max(1,len(v[2:])) for k,v in agg.items()}
    w=out.get('SQ_WAVES',1024.0)
    print(tag, 'waves',int(w), 'per wave per env step:', ' '.join(f"{k[3:]}={v/w/K:.0f}" for k,v in sorted(out.items()) if k!='SQ_WAVES'))
PY
