#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/job3; mkdir -p $o
python -m pytest tests -m gpu -q > $o/tests.log 2>&1; tail -12 $o/tests.log
python tools/ab_libs.py --kernels --reps 2 tools/libocc_gibbs_r3.so tools/libocc_gibbs_nocoop.so occuspytial_amd/libocc_gibbs.so > $o/ab_head.log 2>&1; cat $o/ab_head.log
python tools/ab_libs.py --kernels --reps 2 --lattice 500 500 --chains 1 --iters 300 --warm 60 tools/libocc_gibbs_r3.so tools/libocc_gibbs_nocoop.so occuspytial_amd/libocc_gibbs.so > $o/ab_c4.log 2>&1; cat $o/ab_c4.log
