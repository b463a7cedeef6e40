#!/bin/bash
# GPU (round 6): the whole -m gpu suite + smoke() + the default bench.py line as the driver runs it + the 192-filter tower's
# per-block times (one launch per block / chained), one box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 1300 gpurun_out/bench_line.json; echo
timeout 600 python tools/time_192_chain.py c8,c6 32768 2>&1 | grep -v amdgpu > gpurun_out/r06_time_192_chain.log; tail -13 gpurun_out/r06_time_192_chain.log | cut -c1-160
