#!/bin/bash
# GPU (round 5): the chain through the last block with the heads as its exit (CZ_TOWER_HEADS=1, opt-in): its test + the chain's
# bit-identity test, then per-block times of the 7 x 128 tower, alternating.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_c6.py -m gpu -q -p no:cacheprovider -k "chain" > gpurun_out/pytest_chain.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_chain.log | cut -c1-300
LOG=gpurun_out/ab_chain_heads.log; : > $LOG
for rep in 1 2 3; do
  for c in 0 1; do
    echo "heads_in_chain=$c rep=$rep $(CZ_TOWER_HEADS=$c timeout 200 python tools/time_tower_launches.py c6 32768 masks 2>&1 | grep '^c6')" >> $LOG
  done
done
cat $LOG
