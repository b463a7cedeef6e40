#!/bin/bash
# Round-3 GPU session C: the -m gpu suite (192-filter fused block, rewritten dense tail), the default bench line with
# other_configs, and a kernel trace of a short bench.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
echo "== pytest -m gpu" > gpurun_out/session.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/session.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -15
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" >> gpurun_out/session.log
tail -9 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench.json").readline())
    print("value", round(d["value"]), "sustained", round(d.get("value_sustained") or 0), "ms", round(d["ms_per_step"], 3),
          "search", round(d["roofline_search"]["avg_launch_ms"], 4), "sus", round(d["sustained"]["search_round_ms"], 4), round(d["sustained"]["ms_per_step"], 3),
          "blk", round(d["roofline"]["avg_launch_ms"], 4), "frac", round(d["roofline"]["frac"], 4))
    for k, v in (d.get("other_configs") or {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error", "tree_resets", "compact_queue")}, (v.get("roofline") or {}).get("frac"), (v.get("numerics_check") or {}).get("within_tolerance"))
except Exception as e:
    print("bench parse failed", e)
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_c -o s -- python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist > $ROOT/gpurun_out/prof_c.json 2> $ROOT/gpurun_out/prof_c.err
cd $ROOT
find gpurun_out/prof_c -name '*kernel_trace.csv' -size +20M -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_c/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
cat gpurun_out/session.log
