#!/bin/bash
# A/B of library builds x residual-block schedules inside the normal bench, same box, back to back.
# usage: [CONFIG=deep] bash tools/ab_lib.sh "head:1 new:1 head:0 new:0"   (head -> variants/libczero_head.so, new -> the default build;
#        the number is CZ_RESBLOCK_MODE: 0 = the plain schedules, 1 = the default ones)
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
export TMPDIR=/tmp
for item in $1; do
  lib=${item%%:*}; mode=${item##*:}
  if [ "$lib" = new ]; then unset CZ_LIB; else export CZ_LIB=$PWD/variants/libczero_$lib.so; fi
  CZ_RESBLOCK_MODE=$mode timeout 200 python bench.py --config ${CONFIG:-normal} --steps ${STEPS:-30} --warmup 6 --sustained-rounds 0 --no-micro --no-cpu-baseline --no-other-configs 2>/dev/null > /tmp/ab_item.json
  python - "$item" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_item.json").readline())
r = d["roofline"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), round(r["avg_launch_ms"], 4), flush=True)
PY
done
