#!/bin/bash
# GPU (round 6, session 2): the chains for every arithmetic (csrc/xq_tower.hip): their tests + the network regression files,
# then per-launch times of the 7 x 128 tower per arithmetic, chained (default) and one launch per block, alternating.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_tower.py tests/test_gpu_c6.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_tower.log 2>&1
echo "pytest tower rc=$?"; tail -25 gpurun_out/pytest_tower.log | cut -c1-400
LOG=gpurun_out/r06_ab_chains.log; : > $LOG
for rep in 1 2; do
  for c in 0 1; do
    CZ_TOWER_CHAIN=$c timeout 400 python tools/time_tower_launches.py "c6,c8,c8>3,f16x3" 32768 masks 2>&1 | grep -E "^(c6|c8|f16x3)" | sed "s/^/chain=$c rep=$rep /" >> $LOG
  done
done
cat $LOG
