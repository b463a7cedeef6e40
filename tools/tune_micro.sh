#!/bin/bash
# GPU: the rule micro-suite (bench.micro_suite) at several workgroups-per-CU settings of k_rules_tpb
mkdir -p gpurun_out
: > gpurun_out/tune_micro.log
for v in 8 4 10 12 16; do
  echo "per_cu=$v" >> gpurun_out/tune_micro.log
  CZ_TPB_BLOCKS_PER_CU=$v ITERS=10 timeout 120 python tools/micro_rules.py >> gpurun_out/tune_micro.log 2>&1
done
cat gpurun_out/tune_micro.log
