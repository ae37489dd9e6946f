set -e
mkdir -p gpurun_out/gram
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > gpurun_out/gram/counters.txt 2>&1 || true
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" gpurun_out/gram/counters.txt | sort -u > gpurun_out/gram/mfma_counters.txt || true
cat gpurun_out/gram/mfma_counters.txt
rm -rf gpurun_out/gram/p1 gpurun_out/gram/p2
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/gram/p1 -- python3 tools/rsr_time.py 100 100 1280 4 4 > gpurun_out/gram/p1.log 2> gpurun_out/gram/p1.err
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/gram/p2 -- python3 tools/rsr_time.py 100 100 1280 4 4 > gpurun_out/gram/p2.log 2> gpurun_out/gram/p2.err
python3 - <<'PY'
import csv, glob, collections
for p in ('p1', 'p2'):
    fs = glob.glob('gpurun_out/gram/%s/*/*counter_collection.csv' % p)
    if not fs:
        print(p, 'no counter file'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in agg:
        if 'gram32' in k or 'rsrb_step' in k or 'rsrb_solve' in k:
            print(p, k, {c: '%.4g' % v for c, v in agg[k].items()})
PY
