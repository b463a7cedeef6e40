#!/bin/bash
# GPU: the normal bench (30 timed rounds) with the tower on c8 and on c6, same box, alternating.
# columns: arithmetic | expansions/s | ms per round | ms per residual-block launch | numerics_check | within tolerance
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
export TMPDIR=/tmp
for a in c8 c6 c8 c6; do
  CZ_TOWER_ARITH=$a timeout 200 python bench.py --steps 30 --warmup 6 --sustained-rounds 0 --no-micro --no-cpu-baseline --no-other-configs --no-dist 2>/dev/null > /tmp/w.json
  python - $a <<'PY'
import json, sys
d = json.loads(open("/tmp/w.json").readline())
n = d["numerics_check"]
print("arith", sys.argv[1], d.get("net_arith_effective"), round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["avg_launch_ms"], 4), "logit", n["policy_logit_max_abs_diff"], "policy", n["policy_max_abs_diff"], "value", n["value_max_abs_diff"], n["within_tolerance"], flush=True)
PY
done
