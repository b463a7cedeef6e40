#!/bin/bash
# GPU: A/B of libczero builds on the sustained search probe (tools/search_probe.py), same box, back to back.
#   variants/libczero_<name>.so are built beforehand (git-ignored, they travel with the gpurun snapshot).
mkdir -p gpurun_out
: > gpurun_out/ab_search.log
for rep in ${REPS:-1 2}; do
for f in variants/libczero_*.so; do
  name=$(basename $f .so); name=${name#libczero_}
  echo "variant=$name rep=$rep" >> gpurun_out/ab_search.log
  CZ_LIB=$PWD/$f timeout 300 python tools/search_probe.py --rounds ${ROUNDS:-3000} --timed 200 2>&1 | tail -1 >> gpurun_out/ab_search.log
done
done
cat gpurun_out/ab_search.log
