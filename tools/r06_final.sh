#!/bin/bash
# GPU (round 6, final evidence, ONE box): the default bench.py line exactly as the driver runs it (-> r06_bench_f32.json +
# compact line), then tools/collect_profiles.sh on the same build (kernel trace + stats, PMC passes; summarised on the box with
# per-variant kernel keys and the tower's traffic accounting).  TESTS=1 (default): the whole -m gpu suite + smoke() first.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log | cut -c1-200
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log | cut -c1-200
fi
timeout 900 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cp bench_full.json gpurun_out/r06_bench_f32.json 2>/dev/null
tail -c 1500 gpurun_out/bench_line.json; echo
ROUND=6 timeout 1500 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; echo "collect rc=$?"; tail -4 gpurun_out/collect.log | cut -c1-300
ls gpurun_out/profiles_summary 2>/dev/null
