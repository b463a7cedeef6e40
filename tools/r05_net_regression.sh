#!/bin/bash
# GPU (round 5): regression for a network-only change: the network / guard / drop-in / mask test files, smoke, the driver-like bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_c6.py tests/test_gpu_guard.py tests/test_gpu_dropin.py tests/test_gpu_masks.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_net.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_net.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( unset CZ_BENCH_FULL_LINE; timeout 900 python bench.py > gpurun_out/bench_driver_like.out 2> gpurun_out/bench_driver_like.err )
echo "bench rc=$?"
cp -f bench_full.json gpurun_out/bench_full_final.json 2>/dev/null
python - <<'PY'
import json
line = open("gpurun_out/bench_driver_like.out").read().strip().splitlines()[-1]
d = json.loads(line)
print("compact line bytes:", len(line))
print({k: d.get(k) for k in ("value", "ms_per_step", "value_sustained", "net_arith_effective", "value_peaked_policy", "numerics_peaked_arith", "numerics_logit_max_abs", "numerics_within_tolerance")})
print("roofline:", d.get("roofline")); print("sustained:", d.get("sustained")); print("other:", d.get("other_configs_exp_per_s"))
PY
