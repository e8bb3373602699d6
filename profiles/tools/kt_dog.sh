#!/bin/bash
# per-launch durations of the dogfight kernel over profiles/tools/dog_diag.py (kernel trace), a few windows
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 170 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_dogd -- python $R/profiles/tools/dog_diag.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/kt_dogd/*/*kernel_trace.csv'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'dogfight_env_kernel' in r['Kernel_Name']]
    print(len(d), 'launches')
    for a in (100, 480, 520, 540, 560, 600):
        print(a, ' '.join(f"{x:.0f}" for x in d[a:a+20]))
PY
