set -e
mkdir -p gpurun_out/rsrb2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reduced_rank" > gpurun_out/rsrb2/tests.log 2>&1 || { tail -30 gpurun_out/rsrb2/tests.log; exit 1; }
tail -2 gpurun_out/rsrb2/tests.log
timeout -k 10 200 python tools/rsrb_stamps.py > gpurun_out/rsrb2/stamps.txt 2>&1
cat gpurun_out/rsrb2/stamps.txt
rm -rf gpurun_out/rsrb2/kt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rsrb2/kt -- python3 tools/rsr_time.py 100 100 1280 4 100 > gpurun_out/rsrb2/time.txt 2>gpurun_out/rsrb2/kt.err
cat gpurun_out/rsrb2/time.txt
python3 tools/rsrb_steps.py gpurun_out/rsrb2/kt > gpurun_out/rsrb2/steps.txt
cat gpurun_out/rsrb2/steps.txt
