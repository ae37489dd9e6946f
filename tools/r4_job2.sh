#!/bin/bash
# Developer script (GPU box): GPU tests, PG micro-benchmark, kernel traces of the headline and of config 4.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job2; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; tail -4 $o/tests.log
PG_SD=1.5 rocprofv3 --kernel-trace --output-format csv -d $o/pg -- python tools/pg_occupancy.py > $o/pg.log 2>&1; KSTAT_LIST=1 python tools/kstat.py $o/pg k_draw
rocprofv3 --kernel-trace --output-format csv -d $o/kt_head -- python3 bench.py --no-cpu-baseline > $o/kt_head.json 2> $o/kt_head.err
echo "== headline"; python tools/kstat.py $o/kt_head > $o/kt_head.txt; head -8 $o/kt_head.txt
rocprofv3 --kernel-trace --output-format csv -d $o/kt_c4 -- python3 bench.py --lattice 500 500 --chains-per-gpu 1 --steps 300 --warmup 60 --no-cpu-baseline > $o/kt_c4.json 2> $o/kt_c4.err
echo "== config 4"; python tools/kstat.py $o/kt_c4 > $o/kt_c4.txt; head -8 $o/kt_c4.txt
python - <<PY
import json
for f in ('kt_head','kt_c4'):
    d=json.loads(open('$o/'+f+'.json').read().strip().splitlines()[-1])
    print(f, 'value', round(d['value'],1), 'us/step', round(1e3*d['ms_per_step'],2), 'roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_us','minres_steps_per_launch')}, d['roofline']['dispatch_basis'], d['roofline']['whole_iteration']['frac'])
PY
