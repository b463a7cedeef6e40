#!/bin/bash
# GPU (round 5): the fused input layer's split of its term rounds between the two windows (CZ_FIRST_W1_ROUNDS, default 6) re-tuned
# now that the occupancy boards arrive ready-made: first-launch time of the 7 x 128 c6 tower per setting.
set -u
mkdir -p gpurun_out
LOG=gpurun_out/sweep_first_w1.log; : > $LOG
for rep in 1 2; do
for w in 6 3 4 8 10 12; do
  echo "w1_rounds=$w rep=$rep $(CZ_FIRST_W1_ROUNDS=$w timeout 200 python tools/time_tower_launches.py c6 32768 masks 2>&1 | grep '^c6')" >> $LOG
done
done
cat $LOG
