#!/bin/bash
# GPU (round 6, session 3): the whole -m gpu suite + smoke() on the chained build, the default bench.py line as the driver runs it,
# then the rocprofv3 evidence of the same build (tools/collect_profiles.sh: kernel trace + stats, PMC passes) summarised on the box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cp bench_full.json gpurun_out/bench_full.json 2>/dev/null
tail -c 2500 gpurun_out/bench_line.json
ROUND=6 timeout 1500 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1; echo "collect rc=$?"; tail -5 gpurun_out/collect.log | cut -c1-300
ls gpurun_out/profiles_summary 2>/dev/null
