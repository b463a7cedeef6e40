#!/bin/bash
# GPU: the normal bench (30 timed rounds) on the tower arithmetics (CZ_TOWER_ARITH = bf16x3 | f16x3 | c8), same box, back to back.
# columns: arithmetic | expansions/s | ms per round | ms per residual-block launch | numerics_check (logit / policy / value max abs diff) | within tolerance
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
export TMPDIR=/tmp
for a in bf16x3 f16x3 c8 bf16x3 f16x3 c8; do
  CZ_TOWER_ARITH=$a timeout 200 python bench.py --steps 30 --warmup 6 --sustained-rounds 0 --no-micro --no-cpu-baseline --no-other-configs --no-dist 2>/dev/null > /tmp/w.json
  python - $a <<'PY'
import json, sys
d = json.loads(open("/tmp/w.json").readline())
n = d["numerics_check"]
print("arith", sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["avg_launch_ms"], 4), "logit", n["policy_logit_max_abs_diff"], "policy", n["policy_max_abs_diff"], "value", n["value_max_abs_diff"], n["within_tolerance"], flush=True)
PY
done
