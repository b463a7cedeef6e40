#!/bin/bash
# GPU (round 5): the final regression + the round's committed evidence in one gpurun call (~15 min of box time):
#   the -m gpu suite, smoke, the DRIVER-LIKE bench (plain `python bench.py`; the compact last stdout line is what the driver
#   parses), rocprofv3 kernel stats + PMC passes (tools/collect_profiles.sh, ROUND=5), the search kernels' instruction mix
#   (tools/pmc_valu.sh), the sustained search probe under a kernel trace, single-game latency (tools/uci_nps.py), complete
#   games (tools/measure_games.py), clock / power under the bench, the 11 000-round sustained run.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/pytest_gpu_final.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu_final.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( unset CZ_BENCH_FULL_LINE; timeout 900 python bench.py > gpurun_out/bench_driver_like.out 2> gpurun_out/bench_driver_like.err )
echo "bench rc=$?"
cp -f bench_full.json gpurun_out/bench_full_final.json 2>/dev/null
python - <<'PY'
import json
line = open("gpurun_out/bench_driver_like.out").read().strip().splitlines()[-1]
d = json.loads(line)
print("compact line bytes:", len(line))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_sustained", "net_arith_effective", "value_peaked_policy", "numerics_peaked_arith")})
print("roofline:", d.get("roofline")); print("roofline_search:", d.get("roofline_search")); print("sustained:", d.get("sustained"))
print("other:", d.get("other_configs_exp_per_s")); print("micro:", d.get("micro_suite"))
PY
ROUND=5 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $ROOT; grep "^wrote" gpurun_out/collect.log
if [ "${EXTRAS:-1}" != "1" ]; then exit 0; fi       # EXTRAS=0: regression + bench + profiles only
bash tools/pmc_valu.sh > gpurun_out/pmc_valu.log 2>&1
cd $ROOT
bash tools/profile_search_probe.sh 3000 > gpurun_out/probe_trace.log 2>&1
cd $ROOT; tail -3 gpurun_out/probe_trace.log | cut -c1-600
timeout 200 python tools/uci_nps.py > gpurun_out/uci_nps.log 2>&1; tail -3 gpurun_out/uci_nps.log | cut -c1-300
timeout 400 python tools/measure_games.py --config normal --games 256 > gpurun_out/games.log 2>&1; tail -1 gpurun_out/games.log | cut -c1-500
timeout 120 python tools/clock_power.py --steps 400 > gpurun_out/clock_power.log 2>&1; head -c 500 gpurun_out/clock_power.log; echo
CZ_BENCH_FULL_LINE=1 timeout 900 python bench.py --sustained-rounds 11000 --no-micro --no-cpu-baseline --no-other-configs > gpurun_out/bench_long.json 2> gpurun_out/bench_long.err
tail -3 gpurun_out/bench_long.err | cut -c1-300
