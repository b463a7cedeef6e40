#!/bin/bash
# GPU (round 5): the chained c6 tower (cz_tower_c6): its bit-identity test, then per-launch times of the 7 x 128 tower with the
# inner blocks one per launch (CZ_TOWER_CHAIN=0) and chained (=1), alternating.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_c6.py -m gpu -q -p no:cacheprovider -x -k "chained" > gpurun_out/pytest_chain.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_chain.log | cut -c1-300
LOG=gpurun_out/ab_chain.log; : > $LOG
for rep in 1 2 3; do
  for c in 0 1; do
    echo "chain=$c rep=$rep $(CZ_TOWER_CHAIN=$c timeout 200 python tools/time_tower_launches.py c6 32768 masks 2>&1 | grep '^c6')" >> $LOG
  done
done
cat $LOG
