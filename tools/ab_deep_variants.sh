#!/bin/bash
# GPU: A/B of libczero builds (variants/libczero_<name>.so, tools/build_variant.sh) on the deep leg (20 x 256 fp16), same box, alternating.
mkdir -p gpurun_out
LOG=gpurun_out/ab_deep_variants.log
: > $LOG
for rep in ${REPS:-1 2}; do
  for f in variants/libczero_*.so; do
    name=$(basename $f .so); name=${name#libczero_}
    echo -n "variant=$name rep=$rep " >> $LOG
    CZ_LIB=$PWD/$f timeout 300 python tools/leg_deep.py --one ${SEC:-8} 2>&1 | grep '^{' | tail -1 >> $LOG
  done
done
cat $LOG
