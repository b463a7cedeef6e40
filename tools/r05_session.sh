#!/bin/bash
# GPU (round 5): the round's committed evidence in one gpurun call (~14 min of box time):
#   the -m gpu suite, smoke, the DRIVER-LIKE bench (plain `python bench.py`: the compact last stdout line is what the
#   driver parses; bench_full.json is the full record), rocprofv3 kernel stats + PMC passes of the short bench
#   (tools/collect_profiles.sh), the sustained search probe.     tools/summarize_profiles.py --round 5 afterwards.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
if [ "${SUITE:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -5
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
# exactly what the driver runs: no environment switch, stdout captured, last line parsed
( unset CZ_BENCH_FULL_LINE; timeout 900 python bench.py > gpurun_out/bench_driver_like.out 2> gpurun_out/bench_driver_like.err )
echo "bench rc=$?"
cp -f bench_full.json gpurun_out/bench_full.json 2>/dev/null
python - <<'PY'
import json
line = open("gpurun_out/bench_driver_like.out").read().strip().splitlines()[-1]
d = json.loads(line)
print("compact line bytes:", len(line))
print({k: d.get(k) for k in ("metric", "value", "ms_per_step", "value_sustained", "net_arith_effective", "value_peaked_policy",
                             "numerics_peaked_arith", "n_gpus")})
print("roofline:", d.get("roofline"))
print("cpu_baseline:", d.get("cpu_baseline"))
PY
ROUND=5 bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $ROOT
tail -5 gpurun_out/collect.log
timeout 200 python tools/search_probe.py > gpurun_out/search_probe.json 2> gpurun_out/search_probe.err
head -c 500 gpurun_out/search_probe.json; echo
