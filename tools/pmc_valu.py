"""Developer script: SQ counters per kernel from rocprofv3 --pmc passes, before / after.
    python tools/pmc_valu.py <out.json> <label>=<dir> [<label>=<dir> ...]
Per kernel and label: medians per launch of every counter found, and the derived shares
  valu_busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES-like denominators are not portable; reported instead:
  valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES, issue_share = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES,
  valu_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, parked_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES,
  stalled_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots: the three are disjoint
  parts of a wave's cycles)."""
import csv, glob, json, os, statistics, sys
from collections import defaultdict
out = {'note': 'rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY '
               'SQ_WAIT_INST_ANY, eager launches (OCC_EAGER_ONLY=1: counter collection cannot follow hipGraph launches on ROCm 7.2); '
               'medians per launch', 'runs': {}}
for arg in sys.argv[2:]:
    label, root = arg.split('=', 1)
    files = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
    vals = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if k.startswith('occ::'):
                vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
    run = {}
    for k, cs in sorted(vals.items()):
        m = {c: statistics.median(v) for c, v in cs.items()}
        d = dict(m, launches=len(next(iter(cs.values()))))
        wc = m.get('SQ_WAVE_CYCLES', 0.0)
        if m.get('SQ_WAVES'):
            d['valu_insts_per_wave'] = round(m.get('SQ_INSTS_VALU', 0.0) / m['SQ_WAVES'], 1)
        if wc:
            for name, c in (('issue_share', 'SQ_ACTIVE_INST_ANY'), ('valu_share', 'SQ_ACTIVE_INST_VALU'), ('parked_share', 'SQ_WAIT_ANY'), ('stalled_share', 'SQ_WAIT_INST_ANY')):
                if c in m:
                    d[name] = round(m[c] / wc, 4)
        run[k] = d
    out['runs'][label] = run
json.dump(out, open(sys.argv[1], 'w'), indent=1)
for label, run in out['runs'].items():
    for k in ('occ::k_omega_a<2, 0>', 'occ::k_z_ob<2>', 'occ::k_omega_b<2>'):
        if k in run:
            print(label, k, {a: run[k][a] for a in ('valu_insts_per_wave', 'issue_share', 'valu_share', 'parked_share', 'stalled_share', 'SQ_WAVES') if a in run[k]})
