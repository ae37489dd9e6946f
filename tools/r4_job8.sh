#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job8; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $o/tests.log 2>&1; tail -3 $o/tests.log
for sk in 0 1 2 3; do echo "== tail, OCC_DEBUG_ZOB_SKIP=$sk"; OCC_DEBUG_ZOB_SKIP=$sk timeout -k 10 100 python tools/sizes.py 100,100,4,1500; done
for sk in 0 3; do echo "== OCC_NO_TAIL=1, OCC_DEBUG_ZOB_SKIP=$sk"; OCC_NO_TAIL=1 OCC_DEBUG_ZOB_SKIP=$sk timeout -k 10 100 python tools/sizes.py 100,100,4,1500; done
