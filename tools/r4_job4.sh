#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job4; mkdir -p $o
python -m pytest tests -m gpu -q > $o/tests.log 2>&1; tail -6 $o/tests.log
python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench20.json 2> $o/bench20.err
python bench.py --lattice 500 500 --chains-per-gpu 1 --steps 300 --warmup 60 --no-cpu-baseline > $o/c4.json 2> $o/c4.err
python - <<PY
import json
for f in ('bench','bench20','c4'):
    d=json.loads(open('$o/'+f+'.json').read().strip().splitlines()[-1])
    print(f, 'value', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_us','minres_steps_per_launch')}, d['roofline']['dispatch_basis'], 'whole', d['roofline']['whole_iteration']['frac'], 'inkernel', d['roofline']['in_kernel_clock']['avg_launch_us'])
PY
