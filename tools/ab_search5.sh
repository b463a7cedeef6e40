#!/bin/bash
# GPU (round 5): A/B of the tree kernels on the sustained search probe, same box, alternating:
#   base   = variants/libczero_base.so (the library before this change), planes written
#   new    = csrc/libczero.so, planes written
#   new-m  = csrc/libczero.so, leaves as occupancy boards only (cz_search_leaf_planes(0))
mkdir -p gpurun_out
LOG=gpurun_out/ab_search5.log
: > $LOG
for rep in ${REPS:-1 2}; do
  echo "variant=base rep=$rep" >> $LOG
  CZ_LIB=$PWD/variants/libczero_base.so timeout 300 python tools/search_probe.py --rounds ${ROUNDS:-3000} --timed 200 2>&1 | tail -1 >> $LOG
  echo "variant=new rep=$rep" >> $LOG
  timeout 300 python tools/search_probe.py --rounds ${ROUNDS:-3000} --timed 200 2>&1 | tail -1 >> $LOG
  echo "variant=new-masks-only rep=$rep" >> $LOG
  timeout 300 python tools/search_probe.py --rounds ${ROUNDS:-3000} --timed 200 --masks-only 1 2>&1 | tail -1 >> $LOG
done
cat $LOG
