#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job6; mkdir -p $o
python -m pytest tests -m gpu -q > $o/tests.log 2>&1; tail -5 $o/tests.log
python tools/rsr_time.py 100 100 1280 4 100 > $o/rsrb.log 2>&1; tail -1 $o/rsrb.log
bash tools/profile_round.sh r04 bench pmc c4 valu line > $o/profile.log 2>&1; tail -30 $o/profile.log
