#!/bin/bash
# rocprofv3 kernel trace of a LONG bench run (games in every phase): where a sustained search round's time goes.
#   bash tools/profile_sustained.sh [rounds]     -> gpurun_out/prof_sus/  (summarised by tools/summarize_sustained.py)
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_sus
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=${1:-3000}
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python $ROOT/bench.py --steps 20 --warmup 4 \
    --sustained-rounds $R --no-cpu-baseline --no-micro > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 $ROOT/tools/summarize_sustained.py "$OUT" > "$OUT/summary.json"
find "$OUT" -name '*kernel_trace.csv' -size +10M -delete
cat "$OUT/summary.json" | head -60
