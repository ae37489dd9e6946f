set -e
mkdir -p gpurun_out/rsrb2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/rsrb2/tests.log 2>&1 || { tail -30 gpurun_out/rsrb2/tests.log; exit 1; }
tail -2 gpurun_out/rsrb2/tests.log
timeout -k 10 300 python tools/rsr_time.py 40 50 100 4 2000 > gpurun_out/rsrb2/time.txt 2>&1
timeout -k 10 300 python tools/rsr_time.py 100 100 1280 4 100 >> gpurun_out/rsrb2/time.txt 2>&1
cat gpurun_out/rsrb2/time.txt
