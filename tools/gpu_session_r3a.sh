#!/bin/bash
# Round-3 GPU session A: the whole -m gpu suite on the default build, A/B of the prepared move-generator variants
# (tools/build_variant.sh: quad, quad2, dpp, quad2dpp vs base) on the sustained search probe, the parity suites on the
# variants that win, then the default bench.py line (with other_configs).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
echo "== pytest -m gpu (default build)" > gpurun_out/session.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider --timeout 900 -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/session.log
grep -E "passed|failed|error|deep fp16" gpurun_out/pytest_gpu.log | tail -15
echo "== A/B search probe" >> gpurun_out/session.log
REPS="1 2" ROUNDS=2000 bash tools/ab_search.sh > /dev/null 2>&1
python - <<'PY' | tee -a gpurun_out/session.log
import json, collections
res = collections.defaultdict(list)
name = None
for line in open("gpurun_out/ab_search.log"):
    if line.startswith("variant="):
        name = line.split()[0].split("=")[1]
    elif line.startswith("{") and name:
        res[name].append(json.loads(line)["search_round_ms"]["mean"])
for k, v in sorted(res.items()):
    print("ab", k, [round(x, 4) for x in v])
base = min(res.get("base", [1e9]))
win = [k for k, v in res.items() if k != "base" and v and min(v) < 0.985 * base]
open("gpurun_out/ab_winners.txt", "w").write(" ".join(win))
print("winners:", win)
PY
for v in $(cat gpurun_out/ab_winners.txt); do
  echo "== parity of variant $v" >> gpurun_out/session.log
  CZ_LIB=$ROOT/variants/libczero_$v.so timeout 900 python -m pytest tests/test_gpu_rules.py tests/test_gpu_search.py tests/test_gpu_noise.py -m gpu -q -p no:cacheprovider --timeout 600 > gpurun_out/pytest_$v.log 2>&1
  echo "variant $v pytest rc=$?" >> gpurun_out/session.log
  tail -3 gpurun_out/pytest_$v.log
done
echo "== bench" >> gpurun_out/session.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" >> gpurun_out/session.log
tail -12 gpurun_out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench.json").readline())
    print("value", round(d["value"]), "sustained", round(d.get("value_sustained") or 0), "ms", round(d["ms_per_step"], 2),
          "search", d["roofline_search"]["avg_launch_ms"], "sus search", d["sustained"]["search_round_ms"] if d.get("sustained") else None)
    print("collective", d.get("collective"))
    for k, v in (d.get("other_configs") or {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error", "tree_resets")}, (v.get("roofline") or {}).get("frac"), (v.get("numerics_check") or {}).get("within_tolerance"))
except Exception as e:
    print("bench parse failed", e)
PY
cat gpurun_out/session.log
