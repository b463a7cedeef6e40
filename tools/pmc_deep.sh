#!/bin/bash
# GPU: SQ counters (matrix-pipe busy cycles, waits, LDS activity / conflicts) of the 20 x 256 fp16 network's residual block
# in both schedules: CZ_RESBLOCK_MODE=1 (two channel tiles per matrix wave) and 0 (one).  Output: gpurun_out/pmc_deep_<mode>/
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  OUT=$ROOT/gpurun_out/pmc_deep_$mode
  rm -rf $OUT; mkdir -p $OUT
  CZ_RESBLOCK_MODE=$mode rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      --output-format csv -d $OUT -o p -- python $ROOT/bench.py --config deep --steps 3 --warmup 1 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist > /dev/null 2> $OUT/err.log
  find $OUT -name '*kernel_trace.csv' -size +20M -delete
  python - "$OUT" "$mode" <<'PY'
import csv, glob, sys, collections
out, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_resblock" not in k: continue
        k = k.split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in acc.items():
    n = cnt[k] or 1
    busy = c["SQ_BUSY_CYCLES"]
    print("mode", mode, k, "launches", n, {kk: round(v / n) for kk, v in c.items()})
    if busy: print("   mfma_busy/busy", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, 3), " lds_idx_active/busy", round(c["SQ_LDS_IDX_ACTIVE"] / busy, 3),
                   " lds_conflict/lds_active", round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), 3), " wait_any/wave_cycles", round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3))
PY
done
