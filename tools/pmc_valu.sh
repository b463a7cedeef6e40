#!/bin/bash
# GPU: instruction mix of the search kernels (bench.py, opening-phase rounds): SQ_INSTS_* per launch
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_valu
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/a" -o p -- $BENCH > /dev/null 2> "$OUT/a.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU \
    --output-format csv -d "$OUT/b" -o p -- $BENCH > /dev/null 2> "$OUT/b.err"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = {}
for sub in ("a", "b"):
    fs = glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        out[sub] = "no counter file"; continue
    per = collections.OrderedDict()
    with open(fs[0]) as fh:
        for r in csv.DictReader(fh):
            name = next((n for n in ("k_noise", "k_sim", "k_advance") if n in r["Kernel_Name"]), None)
            if name:
                per.setdefault(int(r["Dispatch_Id"]), {"name": name})[r["Counter_Name"]] = float(r["Counter_Value"])
    seq = [per[i] for i in sorted(per)]
    first = next((i for i, x in enumerate(seq) if x["name"] == "k_sim"), 0)
    seq = seq[first:]
    rounds = [seq[i:i + 4] for i in range(0, len(seq) - len(seq) % 4, 4)]
    rounds = [r for r in rounds if [x["name"] for x in r] == ["k_sim", "k_advance", "k_noise", "k_sim"]][3:]
    labels = ["k_sim(BACKUP)", "k_advance", "k_noise", "k_sim(SELECT)"]
    for j, lab in enumerate(labels):
        acc = collections.defaultdict(float)
        for r in rounds:
            for c, v in r[j].items():
                if c != "name": acc[c] += v / max(1, len(rounds))
        out.setdefault(lab, {}).update(acc)
    out["rounds_" + sub] = len(rounds)
json.dump(out, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find "$OUT" -name '*.csv' -size +8M -delete
