#!/bin/bash
# GPU (round 5): the balanced 192-filter c8 residual block (k_resblock_ip_c8b, CZ_IP_C8_BALANCED=1) against k_resblock_ip_c8.
# The kernel is NOT in the library (measured 14 % slower): `git apply tools/patches/r05_ip_c8_balanced.patch` and rebuild first.
# the 192-filter tests on the new kernel (bit-identical to two cz_conv3x3_c8 launches), then the 10 x 192 leg, alternating.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
CZ_IP_C8_BALANCED=1 timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_guard.py -m gpu -q -p no:cacheprovider -k "192" > gpurun_out/pytest_192.log 2>&1
echo "pytest(192, balanced) rc=$?"; tail -3 gpurun_out/pytest_192.log
LOG=gpurun_out/ab_192.log; : > $LOG
for rep in 1 2; do
  for b in 0 1; do
    echo "balanced=$b rep=$rep" >> $LOG
    CZ_IP_C8_BALANCED=$b timeout 300 python tools/leg_distribute.py 6 c8 2>&1 | grep '^{' >> $LOG
  done
done
cat $LOG
