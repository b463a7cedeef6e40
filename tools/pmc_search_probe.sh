#!/bin/bash
# rocprofv3 PMC pass over tools/search_probe.py (SQ counters only, kernel-trace only): what the waves of the search
# kernels do with their cycles in a sustained state.  Summary: gpurun_out/pmc_probe/summary.json
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_probe
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=${1:-1500}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    --kernel-trace --output-format csv -d "$OUT/sq1" -o p -- python $ROOT/tools/search_probe.py --rounds $R --timed 40 > "$OUT/probe1.json" 2> "$OUT/probe1.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA \
    --kernel-trace --output-format csv -d "$OUT/sq2" -o p -- python $ROOT/tools/search_probe.py --rounds $R --timed 40 > "$OUT/probe2.json" 2> "$OUT/probe2.err"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = {}
for sub in ("sq1", "sq2"):
    fs = glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        out[sub] = "no counter file"; continue
    per = {}
    order = {}
    with open(fs[0]) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            name = next((n for n in ("k_noise", "k_sim", "k_advance") if n in k), None)
            if not name: continue
            did = int(r["Dispatch_Id"])
            order.setdefault(did, name)
            per.setdefault(did, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(order)
    # label launches of a round by position: k_noise, k_sim, k_advance, k_noise, k_sim
    seq = [(order[i], per[i]) for i in ids]
    labels = ["k_noise(B)", "k_sim(BACKUP)", "k_advance", "k_noise(S)", "k_sim(SELECT)"]
    rounds = [seq[i:i + 5] for i in range(0, len(seq) - len(seq) % 5, 5)]
    rounds = [r for r in rounds if [x[0] for x in r] == ["k_noise", "k_sim", "k_advance", "k_noise", "k_sim"]][-200:]
    for j, lab in enumerate(labels):
        acc = {}
        for r in rounds:
            for c, v in r[j][1].items(): acc[c] = acc.get(c, 0.0) + v
        out.setdefault(lab, {}).update({c: v / max(1, len(rounds)) for c, v in acc.items()})
    out["rounds_used_" + sub] = len(rounds)
json.dump(out, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find "$OUT" -name '*.csv' -size +8M -delete
tail -1 "$OUT/probe1.json"
