#!/bin/bash
# One gpurun call = GPU tests + a short bench (+ optional extras given as arguments); everything lands in gpurun_out/.
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" > gpurun_out/session.log
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-10} -p no:cacheprovider --timeout 600 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/session.log
tail -40 gpurun_out/pytest_gpu.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  echo "== bench" >> gpurun_out/session.log
  timeout ${BENCH_TIMEOUT:-600} python bench.py ${BENCH_ARGS:---steps 20 --warmup 4 --sustained-rounds 400} > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench rc=$?" >> gpurun_out/session.log
  tail -5 gpurun_out/bench.err; cut -c1-3000 gpurun_out/bench.json
fi
for extra in "$@"; do
  echo "== $extra" >> gpurun_out/session.log
  timeout ${EXTRA_TIMEOUT:-600} bash -c "$extra" >> gpurun_out/extra.log 2>&1
  echo "rc=$?" >> gpurun_out/session.log
done
cat gpurun_out/session.log
