#!/bin/bash
# GPU: A/B of the rule-code changes: search probe on the variant libraries + the profile build + the micro-suite
mkdir -p gpurun_out
: > gpurun_out/ab_rules.log
for rep in 1 2; do
for name in y2 rules; do
  echo "variant=$name rep=$rep" >> gpurun_out/ab_rules.log
  CZ_LIB=$PWD/variants/libczero_$name.so timeout 300 python tools/search_probe.py --rounds ${ROUNDS:-2000} --timed 200 2>&1 | tail -1 >> gpurun_out/ab_rules.log
done
done
echo "variant=prof" >> gpurun_out/ab_rules.log
CZ_LIB=$PWD/variants/libczero_prof.so timeout 300 python tools/search_probe.py --rounds 3000 --timed 200 2>&1 | tail -1 >> gpurun_out/ab_rules.log
for name in rules tpbf y2; do
  echo "micro=$name" >> gpurun_out/ab_rules.log
  CZ_LIB=$PWD/variants/libczero_$name.so ITERS=10 timeout 120 python tools/micro_rules.py 2>&1 | tail -1 >> gpurun_out/ab_rules.log
done
cat gpurun_out/ab_rules.log
