#!/bin/bash
# GPU (round 6): the ENGINE's rounds (bench.py, compact queue, masks) with cz_tower on k_tower (CZ_TOWER4=0) and on the four-wave
# pair kernel k_resblock_ip4_c8<128> (default), alternating on one box: 300 sustained rounds each (the power-capped state).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp CZ_BENCH_FULL_LINE=1
LOG=gpurun_out/r06_ab_tower4.log; : > $LOG
for rep in 1 2 3; do
  for arith in ${ARITHS:-c6 c8}; do
    for c in 0 1; do
      CZ_TOWER_ARITH=$arith CZ_ARITH_GUARD=0 CZ_TOWER4=$c timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-micro --sustained-rounds 300 --no-other-configs --no-dist 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('arith=$arith tower4=$c rep=$rep', round(d['value']), 'exp/s', round(d['ms_per_step'],3), 'ms/step; sustained', round(d.get('value_sustained') or 0), 'blocks', [round(x,3) for x in r['launch_ms_by_block']][:2])" >> $LOG
    done
  done
done
cat $LOG
