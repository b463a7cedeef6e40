#!/bin/bash
# GPU: the normal bench with the leaves' occupancy boards handed to the fused input layer (default) and without (CZ_LEAF_MASKS=0:
# the first block's copy waves derive them from the planes), same box, alternating.
# columns: CZ_LEAF_MASKS | expansions/s | ms per round | mean ms per residual-block launch | ms of the FIRST launch of a round
export CZ_BENCH_FULL_LINE=1
export TMPDIR=/tmp
for m in 0 1 0 1 0 1; do
  CZ_LEAF_MASKS=$m timeout 200 python bench.py --steps ${STEPS:-60} --warmup 6 --sustained-rounds 0 --no-micro --no-cpu-baseline --no-other-configs --no-dist 2>/dev/null > /tmp/w.json
  python - $m <<'PY'
import json, sys
d = json.loads(open("/tmp/w.json").readline())
print("masks", sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["avg_launch_ms"], 4), d["roofline"].get("first_launch_ms"), flush=True)
PY
done
