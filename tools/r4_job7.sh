#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job7; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1; tail -8 $o/tests.log
echo "== tail"; timeout -k 10 200 python tools/sizes.py 100,100,4,1500 100,100,1,1500 100,100,2,1500 60,60,4,1500
echo "== OCC_NO_TAIL=1"; OCC_NO_TAIL=1 timeout -k 10 200 python tools/sizes.py 100,100,4,1500 100,100,1,1500 100,100,2,1500 60,60,4,1500
