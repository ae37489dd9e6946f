#!/bin/bash
# Developer script (GPU box): the profiles a round commits.  Usage: bash tools/profile_round.sh r02
# Every rocprofv3 command line is echoed into $out/commands.log (and copied to profiles/<tag>_commands.log): the
# flags a number was taken with are part of the evidence.  Counter passes (--pmc) never share a command with a trace
# domain other than --kernel-trace, and run the engine eagerly (OCC_EAGER_ONLY=1: counter collection cannot follow
# hipGraph launches on ROCm 7.2).
set -e
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > $out/commands.log
run() { echo "$*" >> $out/commands.log; "$@"; }
# 1. per-kernel time of the default bench run (headline: 100x100, 4 chains)
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log
cp "$(ls -t $out/stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_bench_kernel_stats.csv
cp "$(ls -t $out/stats/*/*domain_stats.csv | head -n 1)" profiles/${tag}_bench_domain_stats.csv || true
tail -n 1 $out/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof.json
# 2. HBM traffic per kernel: separate counter passes, eager launches
echo "OCC_EAGER_ONLY=1" >> $out/commands.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/fetch.json 2> $out/fetch.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/write.json 2> $out/write.log
python3 tools/pmc_traffic.py $out/fetch $out/write profiles/${tag}_pmc_hbm_traffic.json "100x100 queen lattice, 4 chains" > $out/pmc.log
# 3. BASELINE config 4 (500x500, one chain): bench line, kernel stats, HBM traffic
C4="--lattice 500 500 --chains-per-gpu 1 --steps 300 --warmup 60 --no-cpu-baseline"
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_stats -- python3 bench.py $C4 > $out/c4_under_rocprof.json 2> $out/c4_stats.log
cp "$(ls -t $out/c4_stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_c4_kernel_stats.csv
OCC_EAGER_ONLY=1 run rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c4_fetch -- python3 bench.py --lattice 500 500 --chains-per-gpu 1 --steps 12 --warmup 4 --no-cpu-baseline > $out/c4_fetch.json 2> $out/c4_fetch.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/c4_write -- python3 bench.py --lattice 500 500 --chains-per-gpu 1 --steps 12 --warmup 4 --no-cpu-baseline > $out/c4_write.json 2> $out/c4_write.log
python3 tools/pmc_traffic.py $out/c4_fetch $out/c4_write profiles/${tag}_c4_pmc_hbm_traffic.json "500x500 queen lattice, 1 chains" > $out/c4_pmc.log
python3 bench.py $C4 2> $out/c4_bench.err | tail -n 1 > profiles/${tag}_c4_bench.json
# 4. the reduced-rank sampler (LogitRSRGibbs): kernel stats of ONE profiled run, flags on record (an abort
#    under rocprofv3 in round 1 whose flags had not been recorded)
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/rsr_stats -- python3 tools/rsr_time.py 40 50 100 4 2000 > $out/rsr.log 2> $out/rsr_stats.log
cp "$(ls -t $out/rsr_stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_rsr_kernel_stats.csv
tail -n 1 $out/rsr.log > profiles/${tag}_rsr_bench.txt
# 5. other sizes and paths, one line each (chain-iterations/s; which path every case takes): BASELINE configs 1, 3 (its 8
#    chains on one GPU), 4, 5, the any-placement form and the launch-per-step path on the headline workload
{
  python3 tools/sizes.py 20,20,1,2000 60,60,8,1500 60,60,24,500 100,100,1,1500 100,100,2,1500 100,100,4,2000 100,100,5,1200 100,100,6,1200 100,100,8,1500 100,100,16,800 100,100,32,400 150,150,2,600 250,250,1,400 500,500,1,300
  python3 tools/c5.py 4
  echo "OCC_NO_SCALAR_WAVE=1 (one XCD per chain, eight site waves per workgroup, the first one leads):"; OCC_NO_SCALAR_WAVE=1 python3 tools/sizes.py 100,100,4,1000
  echo "OCC_NO_XCD_LOCAL=1 (any placement):"; OCC_NO_XCD_LOCAL=1 python3 tools/sizes.py 100,100,4,1000
  echo "OCC_NO_PERSISTENT=1 (one launch per MINRES step):"; OCC_NO_PERSISTENT=1 python3 tools/sizes.py 100,100,4,1000
} > profiles/${tag}_sizes.txt 2> $out/sizes.err
# 6. the plain bench line (not profiled), cpu_baseline included
python3 bench.py 2> $out/bench.err | tail -n 1 > profiles/${tag}_bench.json
cp $out/commands.log profiles/${tag}_commands.log
cat profiles/${tag}_bench.json
# the GPU box only returns gpurun_out/: a copy of everything for the caller to move into profiles/
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/
