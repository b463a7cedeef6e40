#!/bin/bash
# A/B of the two (bit-identical) residual-block schedules inside the normal bench: CZ_RESBLOCK_MODE = 0 plain (k_resblock),
# 1 pipelined (k_resblock_pipe).  usage: bash tools/ab_resblock.sh "1 0 1 0"
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
export TMPDIR=/tmp
for v in $1; do
  CZ_RESBLOCK_MODE=$v timeout 200 python bench.py --steps 30 --warmup 6 --sustained-rounds 0 --no-micro --no-cpu-baseline --no-other-configs 2>/dev/null > /tmp/ab_$v.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"/tmp/ab_{v}.json").readline())
r = d["roofline"]
print("mode", v, round(d["value"]), round(d["ms_per_step"], 3), round(r["avg_launch_ms"], 4), flush=True)
PY
done
