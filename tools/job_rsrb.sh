set -e
mkdir -p gpurun_out/rsrb2
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/rsrb2/tests.log 2>&1 || { tail -30 gpurun_out/rsrb2/tests.log; exit 1; }
tail -2 gpurun_out/rsrb2/tests.log
rm -rf gpurun_out/rsrb2/kt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rsrb2/kt -- python3 tools/rsr_time.py 100 100 1280 4 100 > gpurun_out/rsrb2/time.txt 2>gpurun_out/rsrb2/kt.err
python3 tools/rsrb_steps.py gpurun_out/rsrb2/kt > gpurun_out/rsrb2/steps.txt
cat gpurun_out/rsrb2/steps.txt
timeout -k 10 300 python tools/rsr_time.py 100 100 1280 4 100 >> gpurun_out/rsrb2/time.txt 2>&1
timeout -k 10 300 python tools/rsr_time.py 100 100 1300 3 100 >> gpurun_out/rsrb2/time.txt 2>&1
timeout -k 10 300 python tools/rsr_time.py 60 60 468 4 200 >> gpurun_out/rsrb2/time.txt 2>&1
timeout -k 10 300 python tools/rsr_time.py 40 50 100 4 2000 >> gpurun_out/rsrb2/time.txt 2>&1
cat gpurun_out/rsrb2/time.txt
