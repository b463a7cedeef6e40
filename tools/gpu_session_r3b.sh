#!/bin/bash
# Round-3 GPU session B: the -m gpu suite on the new default build (quad generator, single-launch noise, staged input
# convolution, hand-written dense tail), a kernel trace of a short bench, and the A/B of the dense tail.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
echo "== pytest -m gpu" > gpurun_out/session.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/session.log
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -15
B="--steps 20 --warmup 4 --sustained-rounds 300 --no-micro --no-cpu-baseline --no-other-configs"
for mode in 1 0; do
  CZ_FUSED_TAIL=$mode timeout 300 python bench.py $B > gpurun_out/bench_tail$mode.json 2> gpurun_out/bench_tail$mode.err
  python - $mode <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/bench_tail{sys.argv[1]}.json").readline())
print("tail", sys.argv[1], "value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "sus", round(d["value_sustained"]), round(d["sustained"]["ms_per_step"], 3),
      "search", round(d["roofline_search"]["avg_launch_ms"], 4), round(d["sustained"]["search_round_ms"], 4), "blk", round(d["roofline"]["avg_launch_ms"], 4),
      "nc", d["numerics_check"]["policy_logit_max_abs_diff"], d["numerics_check"]["value_max_abs_diff"])
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_b -o s -- python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist > $ROOT/gpurun_out/prof_b.json 2> $ROOT/gpurun_out/prof_b.err
cd $ROOT
find gpurun_out/prof_b -name '*kernel_trace.csv' -size +20M -delete
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_b/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:16]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
cat gpurun_out/session.log
