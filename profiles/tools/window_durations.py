"""profiling tool: the per-step kernel durations of a bench run's last launches, from a rocprofv3 --kernel-trace csv directory."""
import csv, glob, sys
import numpy as np
for d in sys.argv[2:]:
    f = sorted(glob.glob(d + '/*/*kernel_trace.csv'))[-1]
    rows = [r for r in csv.DictReader(open(f)) if 'env_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    k = int(sys.argv[1])
    step = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) < 200000]  # (not the rollout launches)
    x = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in step[-k:]])
    print(d, "last", k, "per-step launches: mean %.2f median %.2f" % (x.mean(), np.median(x)), " ".join("%.1f" % v for v in x))
