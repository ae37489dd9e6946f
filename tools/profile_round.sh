#!/bin/bash
# Developer script (GPU box): the profiles a round commits.  Usage: bash tools/profile_round.sh r03 [part ...]
# (parts: bench pmc c4 valu rsr rsrb sizes line; default: all).  Every rocprofv3 command line is echoed into
# $out/commands.log (and copied to profiles/<tag>_commands.log): the flags a number was taken with are part of the
# evidence.  Counter passes (--pmc) never share a command with a trace domain other than --kernel-trace, and run the
# engine eagerly (OCC_EAGER_ONLY=1: counter collection cannot follow hipGraph launches on ROCm 7.2).
set -e
tag=${1:-r04}
shift || true
parts=${*:-bench pmc c4 valu rsr rsrb sizes line}
out=gpurun_out/prof_$tag
mkdir -p $out profiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
touch $out/commands.log
run() { echo "$*" >> $out/commands.log; "$@"; }
has() { case " $parts " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
# 1. per-kernel time of the default bench run (headline: 100x100, 4 chains); the k_iter dispatches of the TIMED REGION from
#    the same trace (what roofline.frac's dispatch-duration basis is checked against)
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.log
cp "$(ls -t $out/stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_bench_kernel_stats.csv
cp "$(ls -t $out/stats/*/*domain_stats.csv | head -n 1)" profiles/${tag}_bench_domain_stats.csv || true
tail -n 1 $out/bench_under_rocprof.json > profiles/${tag}_bench_under_rocprof.json
python3 tools/trace_region.py "$(ls -t $out/stats/*/*kernel_trace.csv | head -n 1)" 200 2000 profiles/${tag}_bench_trace_region.json > $out/trace_region.log
# ... and of the driver's own short call
run rocprofv3 --kernel-trace --output-format csv -d $out/stats20 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20_under_rocprof.json 2> $out/stats20.log
python3 tools/trace_region.py "$(ls -t $out/stats20/*/*kernel_trace.csv | head -n 1)" 5 20 profiles/${tag}_bench_20_steps_trace_region.json > $out/trace_region20.log
fi
if has pmc; then
# 2. HBM traffic per kernel: separate counter passes, eager launches
echo "OCC_EAGER_ONLY=1" >> $out/commands.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/fetch.json 2> $out/fetch.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline > $out/write.json 2> $out/write.log
python3 tools/pmc_traffic.py $out/fetch $out/write profiles/${tag}_pmc_hbm_traffic.json "100x100 queen lattice, 4 chains" > $out/pmc.log
fi
if has c4; then
# 3. BASELINE config 4 (500x500, one chain; k_tiles): bench line, kernel stats + timed region, HBM traffic
C4="--lattice 500 500 --chains-per-gpu 1 --steps 300 --warmup 60 --no-cpu-baseline"
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_stats -- python3 bench.py $C4 > $out/c4_under_rocprof.json 2> $out/c4_stats.log
cp "$(ls -t $out/c4_stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_c4_kernel_stats.csv
python3 tools/trace_region.py "$(ls -t $out/c4_stats/*/*kernel_trace.csv | head -n 1)" 60 300 profiles/${tag}_c4_trace_region.json k_tiles > $out/c4_trace_region.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c4_fetch -- python3 bench.py --lattice 500 500 --chains-per-gpu 1 --steps 12 --warmup 4 --no-cpu-baseline > $out/c4_fetch.json 2> $out/c4_fetch.log
OCC_EAGER_ONLY=1 run rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/c4_write -- python3 bench.py --lattice 500 500 --chains-per-gpu 1 --steps 12 --warmup 4 --no-cpu-baseline > $out/c4_write.json 2> $out/c4_write.log
python3 tools/pmc_traffic.py $out/c4_fetch $out/c4_write profiles/${tag}_c4_pmc_hbm_traffic.json "500x500 queen lattice, 1 chains" > $out/c4_pmc.log
python3 bench.py $C4 2> $out/c4_bench.err | tail -n 1 > profiles/${tag}_c4_bench.json
OCC_NO_TILES=1 python3 bench.py $C4 2> $out/c4_lps.err | tail -n 1 > profiles/${tag}_c4_bench_launch_per_step.json
fi
if has valu; then
# 3b. the Polya-Gamma kernels, SQ counters, BEFORE (round 3's build, tools/libocc_gibbs_r3.so: built by hand from commit c4e04c1,
#     see DESIGN 4) and AFTER, config 4 and the headline, eager launches
SQC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
echo "OCC_EAGER_ONLY=1 [OCC_LIB=tools/libocc_gibbs_r3.so]" >> $out/commands.log
if [ -f tools/libocc_gibbs_r3.so ]; then
OCC_LIB=tools/libocc_gibbs_r3.so OCC_EAGER_ONLY=1 run rocprofv3 --pmc $SQC --output-format csv -d $out/valu_c4_before -- python3 tools/pmc_run.py 500 500 1 10 > $out/valu_c4_before.log 2>&1
OCC_LIB=tools/libocc_gibbs_r3.so OCC_EAGER_ONLY=1 run rocprofv3 --pmc $SQC --output-format csv -d $out/valu_head_before -- python3 tools/pmc_run.py 100 100 4 30 > $out/valu_head_before.log 2>&1
fi
OCC_EAGER_ONLY=1 run rocprofv3 --pmc $SQC --output-format csv -d $out/valu_c4_after -- python3 tools/pmc_run.py 500 500 1 10 > $out/valu_c4_after.log 2>&1
OCC_EAGER_ONLY=1 run rocprofv3 --pmc $SQC --output-format csv -d $out/valu_head_after -- python3 tools/pmc_run.py 100 100 4 30 > $out/valu_head_after.log 2>&1
python3 tools/pmc_valu.py profiles/${tag}_pmc_valu.json c4_before=$out/valu_c4_before c4_after=$out/valu_c4_after headline_before=$out/valu_head_before headline_after=$out/valu_head_after > $out/valu.log 2>&1
cat $out/valu.log
fi
if has rsr; then
# 4. the reduced-rank sampler (LogitRSRGibbs), LDS-resident solve (m = 100)
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/rsr_stats -- python3 tools/rsr_time.py 40 50 100 4 2000 > $out/rsr.log 2> $out/rsr_stats.log
cp "$(ls -t $out/rsr_stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_rsr_kernel_stats.csv
tail -n 1 $out/rsr.log > profiles/${tag}_rsr_bench.txt
fi
if has rsrb; then
# 4b. ... and with the reference's default threshold at 100x100: 1 280 basis columns, the device-memory solve (k_rsrb_*)
run rocprofv3 --kernel-trace --stats --output-format csv -d $out/rsrb_stats -- python3 tools/rsr_time.py 100 100 1280 4 100 > $out/rsrb.log 2> $out/rsrb_stats.log
cp "$(ls -t $out/rsrb_stats/*/*kernel_stats.csv | head -n 1)" profiles/${tag}_rsrb_kernel_stats.csv
{ echo "under rocprofv3 --kernel-trace:"; tail -n 1 $out/rsrb.log; echo "not profiled:"; python3 tools/rsr_time.py 100 100 1280 4 100 | tail -n 1; python3 tools/rsr_time.py 100 100 1280 3 100 | tail -n 1; } > profiles/${tag}_rsrb_bench.txt
fi
if has sizes; then
# 5. other sizes and paths, one line each (chain-iterations/s; which path every case takes)
{
  python3 tools/sizes.py 20,20,1,2000 60,60,8,1500 60,60,24,500 100,100,1,1500 100,100,2,1500 100,100,4,2000 100,100,5,1200 100,100,6,1200 100,100,8,1500 100,100,16,800 100,100,32,400 150,150,2,600 250,250,1,400 250,250,2,300 350,350,1,300 250,250,4,200 500,500,1,300
  python3 tools/c5.py 4
  echo "OCC_NO_SCALAR_WAVE=1 (one XCD per chain, eight site waves per workgroup, the first one leads):"; OCC_NO_SCALAR_WAVE=1 python3 tools/sizes.py 100,100,4,1000
  echo "OCC_NO_XCD_LOCAL=1 (any placement):"; OCC_NO_XCD_LOCAL=1 python3 tools/sizes.py 100,100,4,1000 250,250,1,400
  echo "OCC_NO_TILES=1 (k_iter any placement / one launch per MINRES step where k_tiles is the default):"; OCC_NO_TILES=1 python3 tools/sizes.py 250,250,1,400 250,250,2,300 350,350,1,300 500,500,1,300
  echo "OCC_NO_PERSISTENT=1 (one launch per MINRES step):"; OCC_NO_PERSISTENT=1 python3 tools/sizes.py 100,100,4,1000
} > profiles/${tag}_sizes.txt 2> $out/sizes.err
fi
if has line; then
# 6. the plain bench lines (not profiled): default (cpu_baseline included) and the driver's short call
python3 bench.py 2> $out/bench.err | tail -n 1 > profiles/${tag}_bench.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $out/bench20.err | tail -n 1 > profiles/${tag}_bench_20_steps.json
cat profiles/${tag}_bench.json
fi
cp $out/commands.log profiles/${tag}_commands.log
# the GPU box only returns gpurun_out/: a copy of everything for the caller to move into profiles/
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/
