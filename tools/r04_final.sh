#!/bin/bash
# round 4, final regression: the whole -m gpu suite, smoke(), the default bench line
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_final.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/pytest_gpu_final.log
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
keys = ("value", "ms_per_step", "value_sustained", "roofline_frac", "roofline_frac_sustained", "net_arith_requested", "net_arith_effective",
        "numerics_logit_max_abs", "numerics_peaked_policy_max_abs", "numerics_peaked_arith", "games_per_hour_steady_state")
print({k: d.get(k) for k in keys})
print({k: round(v["value"]) for k, v in d.get("other_configs", {}).items() if isinstance(v, dict) and "value" in v})
PY
