#!/bin/bash
# Developer script (GPU box): the profiles a round commits.  Usage: bash tools/profile_round.sh r01
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. per-kernel time of the default bench run
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log
cp "$(ls -t $out/stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_bench_kernel_stats.csv
cp "$(ls -t $out/stats/*/*domain_stats.csv | head -n 1)" profiles/${tag}_bench_domain_stats.csv || true
tail -n 1 $out/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof.json
# 2. HBM traffic per kernel: separate counter passes, eager launches
OCC_EAGER_ONLY=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/fetch.json 2> $out/fetch.log
OCC_EAGER_ONLY=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/write.json 2> $out/write.log
python tools/pmc_traffic.py $out/fetch $out/write profiles/${tag}_pmc_hbm_traffic.json "100x100 queen lattice, 4 chains" > $out/pmc.log
# 3. the plain bench line (not profiled)
python bench.py 2> $out/bench.err | tail -n 1 > profiles/${tag}_bench.json
cat profiles/${tag}_bench.json
# the GPU box only returns gpurun_out/: a copy of everything for the caller to move into profiles/
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/
