#!/bin/bash
# Quick sanity of the non-default bench configurations (each a few rounds): prints value / ms per step / numerics.
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py "$@" --steps 6 --warmup 2 --sustained-rounds 0 --no-micro --no-cpu-baseline 2>/tmp/err.log > /tmp/out.json || { echo "FAILED: $*"; tail -5 /tmp/err.log; return; }
  python - "$*" <<'PY'
import json, sys
d = json.loads(open("/tmp/out.json").readline())
nc = d.get("numerics_check") or {}
print(sys.argv[1], "->", round(d["value"]), "exp/s", round(d["ms_per_step"], 2), "ms", "compact", d["config"].get("compact_queue"),
      "policy diff", nc.get("policy_max_abs_diff"), "value diff", nc.get("value_max_abs_diff"), flush=True)
PY
}
run --graph
run --dtype bfloat16
run --dtype float16
run --trunk library
run --config mini
run --config deep --games 1024
run --sims-per-round 16 --games 2048
