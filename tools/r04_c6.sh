#!/bin/bash
# round 4: the c6 arithmetic -- tests, then the bench with the arithmetic requested
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_c6.py -x -q -s > gpurun_out/c6_tests.log 2>&1
echo "c6 tests rc=$?"; grep -v "amdgpu.ids" gpurun_out/c6_tests.log | tail -25
timeout 900 python bench.py --arith c6 > gpurun_out/bench_c6.json 2> gpurun_out/bench_c6.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_c6.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_c6.json").read().strip().splitlines()[-1])
keys = ("value", "ms_per_step", "value_sustained", "roofline_frac", "net_arith_requested", "net_arith_effective",
        "numerics_logit_max_abs", "numerics_peaked_policy_max_abs", "numerics_peaked_arith")
print({k: d.get(k) for k in keys})
print(d.get("roofline"))
PY
