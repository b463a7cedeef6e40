#!/bin/bash
# GPU (round 6): the guard walking the hybrids one block at a time -- the tests that look at its candidates, then the default
# bench line (its peaked_policy leg runs whatever the guard now picks).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_guard.py tests/test_gpu_c6.py tests/test_gpu_tower.py tests/test_gpu_dropin.py -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_guard.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_guard.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_line_guard.json 2> gpurun_out/bench_guard.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/bench_line_guard.json; echo
grep -o '"peaked_policy": {[^}]*}' gpurun_out/bench_guard.err | head -2 | cut -c1-1500
