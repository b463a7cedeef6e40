#!/bin/bash
# GPU: everything the round's committed profiles come from, in one call:
#   the -m gpu suite, rocprofv3 kernel stats + PMC passes of the short bench (tools/collect_profiles.sh), the instruction
#   mix of the search kernels (tools/pmc_valu.sh), the default bench line (with other_configs and the CPU baseline), the
#   kernel trace of the sustained search probe, complete games (games/hour conversion), clock / power under the bench,
#   the A/B of the tower arithmetics, a long sustained run.
#   tools/summarize_profiles.py --round N turns gpurun_out/prof into profiles/rNN_*.
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $ROOT
bash tools/pmc_valu.sh > gpurun_out/pmc_valu.log 2>&1
cd $ROOT
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -12 gpurun_out/bench_default.err
bash tools/profile_search_probe.sh 3000 > gpurun_out/probe_trace.log 2>&1
tail -3 gpurun_out/probe_trace.log
cd $ROOT
timeout 300 python tools/measure_games.py --config normal --games 256 > gpurun_out/games.log 2>&1
tail -2 gpurun_out/games.log
# shader clock / socket power under the tower (is it at the power cap?) and the A/B of the two tower arithmetics
timeout 120 python tools/clock_power.py --steps 400 > gpurun_out/clock_power.log 2>&1
head -c 400 gpurun_out/clock_power.log; echo
bash tools/ab_arith.sh > gpurun_out/ab_arith.log 2>&1
cat gpurun_out/ab_arith.log
if [ "${LONG:-1}" = "1" ]; then
  timeout 900 python bench.py --sustained-rounds 11000 --no-micro --no-cpu-baseline --no-other-configs > gpurun_out/bench_long.json 2> gpurun_out/bench_long.err
  tail -3 gpurun_out/bench_long.err
fi
