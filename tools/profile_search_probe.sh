#!/bin/bash
# rocprofv3 kernel trace of tools/search_probe.py: per-kernel times of the search round in a sustained state.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_probe
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python $ROOT/tools/search_probe.py --rounds ${1:-3000} > "$OUT/probe.json" 2> "$OUT/probe.err"
python3 $ROOT/tools/summarize_sustained.py "$OUT" > "$OUT/summary.json" 2>> "$OUT/probe.err"
find "$OUT" -name '*kernel_trace.csv' -size +10M -delete
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
w = d["last_1000_rounds"]
print({k: (round(v["mean"], 1), round(v["p50"], 1), round(v["p99"], 1)) if isinstance(v, dict) else round(v, 1) for k, v in w.items()})
PY
tail -1 "$OUT/probe.json"
