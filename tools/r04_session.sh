#!/bin/bash
# GPU (round 4): everything the round's committed profiles come from, in one gpurun call (~17 min of box time):
#   the -m gpu suite, smoke, rocprofv3 kernel stats + PMC passes of the short bench (tools/collect_profiles.sh), the default
#   bench line, the A/B of the tower arithmetics, the c8 K-loop probe (shader cycles, clock), the in-kernel section stamps
#   of k_resblock_c8 (variants/libczero_stamps.so, built beforehand), launch times of the block kernels, clock / power
#   under the bench, complete games, the 11 000-round sustained run.   tools/summarize_profiles.py --round 4 afterwards.
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $ROOT
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -14 gpurun_out/bench_default.err | grep bench
bash tools/ab_arith.sh > gpurun_out/ab_arith.log 2>&1
cat gpurun_out/ab_arith.log
timeout 60 tools/probes/c8_kloop_probe 1.0 2>&1 | grep RESULT > gpurun_out/c8_kloop_probe.log
cat gpurun_out/c8_kloop_probe.log
CZ_LIB=variants/libczero_stamps.so timeout 120 python tools/rb_stamps.py > gpurun_out/rb_stamps.json 2> gpurun_out/rb_stamps.err
head -c 600 gpurun_out/rb_stamps.json; echo
timeout 120 python tools/time_resblock_c8.py > gpurun_out/time_resblock_c8.json 2>&1
tail -1 gpurun_out/time_resblock_c8.json | head -c 600; echo
timeout 120 python tools/clock_power.py --steps 400 > gpurun_out/clock_power.log 2>&1
head -c 300 gpurun_out/clock_power.log; echo
timeout 300 python tools/measure_games.py --config normal --games 256 > gpurun_out/games.log 2>&1
tail -2 gpurun_out/games.log | head -c 600; echo
if [ "${LONG:-1}" = "1" ]; then
  timeout 900 python bench.py --sustained-rounds 11000 --no-micro --no-cpu-baseline --no-other-configs > gpurun_out/bench_long.json 2> gpurun_out/bench_long.err
  tail -3 gpurun_out/bench_long.err
fi
