#!/bin/bash
# quick GPU check: the network test files + a kernel trace of a short bench (per-kernel averages)
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_dropin.py -m gpu -q --maxfail=10 -p no:cacheprovider --timeout 600 ${PYTEST_K:-} > gpurun_out/pytest_quick.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_quick.log | tail -8
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_q -o s -- python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist > $ROOT/gpurun_out/prof_q.json 2> $ROOT/gpurun_out/prof_q.err
cd $ROOT
find gpurun_out/prof_q -name '*kernel_trace.csv' -size +20M -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_q/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
