"""Developer script: HBM bytes per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> "<workload>"

Counter values are KB per dispatch.  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE
reads half of what a wide coalesced stream fetches (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, glob, json, os, statistics, sys
from collections import defaultdict


def per_kernel(root, counter):
    f = max(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
    vals = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            vals[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return vals


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `OCC_EAGER_ONLY=1 python bench.py '
               '--steps 40 --warmup 10 --no-cpu-baseline` (eager launches of the same kernels: counter collection '
               'crashes on hipGraph launches in ROCm 7.2). Values are medians per launch in KB as reported; hbm_bytes '
               '= (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md '
               '(x2 for wide coalesced reads).',
       'workload': sys.argv[4], 'kernels': {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith('occ::'):
        continue
    fk = statistics.median(fetch.get(k, [0.0]))
    wk = statistics.median(write.get(k, [0.0]))
    out['kernels'][k] = {'FETCH_SIZE_KB': fk, 'WRITE_SIZE_KB': wk, 'launches': len(fetch.get(k, [])),
                         'hbm_bytes_per_launch': int((2 * fk + wk) * 1024)}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1))
