#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline figures (run ON THE GPU BOX, e.g. through gpurun):
#   1. --kernel-trace --stats of the default bench.py run                      -> gpurun_out/prof/stats
#   2. PMC passes (one per counter set, never combined with trace domains other than kernel-trace):
#        SQ set (MFMA busy, wave cycles, waits, LDS)   FETCH_SIZE   WRITE_SIZE
# tools/summarize_profiles.py turns gpurun_out/prof into profiles/rNN_*.{json,md}.
export CZ_BENCH_FULL_LINE=1   # bench.py prints its full record on stdout for these scripts (round 5: the default is the compact line)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $BENCH > "$OUT/stats.json" 2> "$OUT/stats.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_grbm.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
# summarise on the box (the counter tables of one pass exceed what gpurun carries back) and keep the payload small
python $ROOT/tools/summarize_profiles.py --round ${ROUND:-4} --src "$OUT" --out $ROOT/gpurun_out/profiles_summary
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
ls -la "$OUT" "$OUT"/*/ 2>/dev/null | head -40
