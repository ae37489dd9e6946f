"""Developer script: per-kernel mean / min / max duration (us) and count from a rocprofv3 --kernel-trace csv.
    python tools/kstat.py <dir or csv> [name-filter]"""
import csv, glob, os, sys
from collections import defaultdict
path = sys.argv[1]
files = [path] if path.endswith('.csv') else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
d = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if flt in r['Kernel_Name']:
            d[r['Kernel_Name']].append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    v.sort()
    us = [x[1] for x in v]
    print('%-60s n %6d  mean %9.2f  min %9.2f  max %9.2f' % (k[:60], len(us), sum(us) / len(us), min(us), max(us)))
    if os.environ.get('KSTAT_LIST'):
        print('   in launch order:', ' '.join('%.1f' % u for u in us[:64]))
