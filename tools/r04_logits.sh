#!/bin/bash
# round 4: the engine queue as raw logits (cz_search_policy_logits) -- tests, then bench with and without
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_search.py tests/test_gpu_conv.py -x -q -m gpu -k "logit or heads_tail or compact" > gpurun_out/logits_tests.log 2>&1
echo "new tests rc=$?"; tail -3 gpurun_out/logits_tests.log
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_logits.log 2>&1
echo "suite rc=$?"; tail -3 gpurun_out/pytest_gpu_logits.log
python bench.py > gpurun_out/bench_logits.json 2> gpurun_out/bench_logits.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_logits.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_sustained", "roofline_frac")}, d["config"].get("policy_rows"))
PY
