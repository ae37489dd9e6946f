"""Developer script: mean duration of the k_iter dispatches of bench.py's TIMED REGION from a rocprofv3 kernel trace of
the same command (profiles/rNN_bench_trace_region.json).  Usage: trace_region.py <kernel_trace.csv> <warmup> <steps> <out.json>
The k_iter launches of a bench run, in order: the residency probes at creation (a few microseconds each), `warmup`
iterations, `steps` timed iterations, 200 profile-pass iterations."""
import csv
import json
import sys

import numpy as np

path, warmup, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
kern = sys.argv[5] if len(sys.argv) > 5 else 'k_iter'
rows = [r for r in csv.DictReader(open(path)) if ('occ::' + kern + '<') in r['Kernel_Name']]
d = np.array([int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]) / 1e3
probes = int(np.sum(d[:12] < 10.0))                       # the residency probes return within microseconds
lo, hi = probes + warmup, probes + warmup + steps
res = {'source': path.split('/')[-1], 'kernel': rows[0]['Kernel_Name'], 'k_iter_dispatches': len(d), 'residency_probes': probes,
       'warmup': warmup, 'steps': steps, 'timed_region_mean_us': round(float(d[lo:hi].mean()), 3),
       'timed_region_min_us': round(float(d[lo:hi].min()), 3), 'timed_region_max_us': round(float(d[lo:hi].max()), 3),
       'warmup_mean_us': round(float(d[probes:lo].mean()), 3) if warmup else None,
       'after_region_mean_us': round(float(d[hi:].mean()), 3) if len(d) > hi else None,
       'all_dispatches_mean_us': round(float(d.mean()), 3)}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
