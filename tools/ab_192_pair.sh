#!/bin/bash
# GPU: the 192-filter chain with one board (CZ_IP_PAIR=0) and a pair of boards (default) per workgroup: bit-identity tests, then the
# tower's per-block times, same box, alternating.
mkdir -p gpurun_out
LOG=gpurun_out/ab_192_pair.log
: > $LOG
timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_c6.py tests/test_gpu_guard.py -m gpu -q -x -p no:cacheprovider -k "192" 2>&1 | tail -3 | tee -a $LOG
for rep in 1 2; do
  for v in 0 1; do
    echo "pair=$v rep=$rep" >> $LOG
    CZ_IP_PAIR=$v timeout 300 python tools/time_192_chain.py ${ARITHS:-c8,c6} 32768 2>&1 | grep -E "chain|blocks" | grep -v same | grep -v "^{" >> $LOG
  done
done
cat $LOG | cut -c1-160
