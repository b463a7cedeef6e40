#!/bin/bash
# GPU (round 6): the ENGINE's rounds (bench.py, compact queue, masks) per tower arithmetic, chained (default) against one launch per
# block (CZ_TOWER_CHAIN=0), alternating on one box: expansions/s, ms per step, per-block times of the tower.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp CZ_BENCH_FULL_LINE=1
LOG=gpurun_out/r06_ab_engine_chain.log; : > $LOG
for rep in 1 2; do
  for arith in c6 c8 f16x3; do
    for c in 0 1; do
      CZ_TOWER_ARITH=$arith CZ_ARITH_GUARD=0 CZ_TOWER_CHAIN=$c timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-micro --sustained-rounds 0 --no-other-configs --no-dist 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('arith=$arith chain=$c rep=$rep', round(d['value']), 'exp/s', round(d['ms_per_step'],3), 'ms/step blocks', [round(x,3) for x in r['launch_ms_by_block']], r['kernel_short'][:90])" >> $LOG
    done
  done
done
cat $LOG
