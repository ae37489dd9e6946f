"""Developer script: per-iteration timeline from a rocprofv3 --kernel-trace CSV (gaps between kernels).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python tools/steady.py 600
    python tools/gaps.py gpurun_out/kt
"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
f = max(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.split('(')[0].replace('occ::', '').replace('void ', '')
    return n.split('<')[0]
ev = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '')) for r in rows]
# last 200 iterations: anchor on k_z_ob
zi = [i for i, e in enumerate(ev) if e[0] == 'k_z_ob']
zi = zi[-min(201, len(zi)):]
dur = defaultdict(list); gap_after = defaultdict(list); per_it = []
for a, b in zip(zi[:-1], zi[1:]):
    seq = ev[a + 1:b + 1]
    per_it.append(ev[b][2] - ev[a][2])
    main = [e for e in seq if e[0] in ('k_eta_init', 'k_solve', 'k_iter', 'k_minres', 'k_beta_partial', 'k_z_ob')]
    prev_end = ev[a][2]; prev = 'k_z_ob(prev)'
    for e in main:
        dur[e[0]].append(e[2] - e[1])
        gap_after[prev + ' -> ' + e[0]].append(e[1] - prev_end)
        prev_end = e[2]; prev = e[0]
    for e in seq:
        if e[0] in ('k_omega_a', 'k_alpha_draw', 'k_noise'):
            dur[e[0]].append(e[2] - e[1])
print('iterations analysed: %d, mean %.1f us per iteration (end of k_z_ob to end of k_z_ob)' % (len(per_it), sum(per_it) / len(per_it) / 1e3))
print('kernel durations (us): mean  [count per iteration]')
for k, v in dur.items():
    print('  %-16s %8.2f   x%.2f' % (k, sum(v) / len(v) / 1e3, len(v) / len(per_it)))
print('gaps on the critical path (us):')
for k, v in gap_after.items():
    print('  %-36s %8.2f   x%.2f' % (k, sum(v) / len(v) / 1e3, len(v) / len(per_it)))
