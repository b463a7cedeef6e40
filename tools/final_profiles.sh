#!/bin/bash
# GPU: everything the round's committed profiles come from, in one call:
#   rocprofv3 kernel stats + PMC passes of the short bench (tools/collect_profiles.sh), the instruction mix of the search
#   kernels (tools/pmc_valu.sh), the default bench line (with other_configs and the CPU baseline), the kernel trace of
#   the sustained search probe.  tools/summarize_profiles.py --round N turns gpurun_out/prof into profiles/rNN_*.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $ROOT
bash tools/pmc_valu.sh > gpurun_out/pmc_valu.log 2>&1
cd $ROOT
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -12 gpurun_out/bench_default.err
bash tools/profile_search_probe.sh 3000 > gpurun_out/probe_trace.log 2>&1
tail -3 gpurun_out/probe_trace.log
ls gpurun_out/prof gpurun_out/prof/*/ | head -40
