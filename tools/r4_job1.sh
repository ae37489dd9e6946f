#!/bin/bash
# Developer script (GPU box): round-4 checkpoint -- GPU tests, then kernel traces of the headline and of config 4.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job1; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; tail -4 $o/tests.log
rocprofv3 --kernel-trace --output-format csv -d $o/kt_head -- python3 bench.py --no-cpu-baseline > $o/kt_head.json 2> $o/kt_head.err
echo "== headline"; python tools/kstat.py $o/kt_head | head -12
for gb in 1 2; do
  OCC_TILES_GB=$gb rocprofv3 --kernel-trace --output-format csv -d $o/kt_c4_gb$gb -- python3 bench.py --lattice 500 500 --chains-per-gpu 1 --steps 300 --warmup 60 --no-cpu-baseline > $o/kt_c4_gb$gb.json 2> $o/kt_c4_gb$gb.err
  echo "== config 4, GB=$gb"; python tools/kstat.py $o/kt_c4_gb$gb | head -8
  python - <<PY
import json
d=json.loads(open('$o/kt_c4_gb$gb.json').read().strip().splitlines()[-1])
print('c4 gb$gb value', d['value'], 'ms/step', d['ms_per_step'], 'k_tiles us', d['roofline']['in_kernel_clock']['avg_launch_us'])
PY
done
python - <<PY
import json
d=json.loads(open('$o/kt_head.json').read().strip().splitlines()[-1])
print('headline value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_us','minres_steps_per_launch')}, d['roofline']['dispatch_basis'])
PY
