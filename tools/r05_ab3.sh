#!/bin/bash
# GPU (round 5, third A/B): the input-layer table with contiguous 16-byte loads + DPP row sums in the fused head convolutions
# (new) against the library before them (variants/libczero_r5b.so): network tests on the new library, then per-launch times
# of the 7 x 128 tower, alternating.  (The table layout belongs to the library: each run builds its table with its own
# _native.input_table -- the variant is run from a checkout of its own commit, variants/r5b_tree.)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_masks.py tests/test_gpu_c6.py tests/test_gpu_dropin.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_net.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_net.log
LOG=gpurun_out/ab_first_table.log; : > $LOG
for rep in 1 2 3; do
  echo "variant=before rep=$rep" >> $LOG
  ( cd variants/r5b_tree && timeout 200 python tools/time_tower_launches.py c6 32768 masks 2>&1 | grep "^c6" >> ../../$LOG )
  echo "variant=new rep=$rep" >> $LOG
  timeout 200 python tools/time_tower_launches.py c6 32768 masks 2>&1 | grep "^c6" >> $LOG
done
cat $LOG
