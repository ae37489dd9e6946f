"""Developer script: mean duration of every k_rsrb_step launch of an iteration, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/rsr_time.py 100 100 1280 4 100
    python tools/rsrb_steps.py DIR"""
import collections
import csv
import glob
import sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = collections.defaultdict(list)
tot = collections.defaultdict(float)
cnt = collections.Counter()
i = 0
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = n.split('(')[0].replace('void ', '')
    tot[key] += d
    cnt[key] += 1
    if 'k_rsrb_assemble' in n:
        i = 0
    if 'k_rsrb_step' in n:
        per[i].append(d)
        i += 1
print('k_rsrb_step, launch 0 (the head) .. last, mean us:')
print(' '.join('%.1f' % (sum(per[k]) / len(per[k])) for k in sorted(per)))
its = max(cnt.get('occ::k_rsrb_solve', 1), 1)
for k in sorted(tot, key=lambda k: -tot[k])[:10]:
    print('%-40s %8.1f us per iteration (%d launches)' % (k, tot[k] / its, cnt[k] // its))
