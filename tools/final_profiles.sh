#!/bin/bash
# GPU: everything the round's committed profiles come from, in one call.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$(pwd)
bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1
cd $ROOT
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/bench_default.err
bash tools/profile_search_probe.sh 3000 > gpurun_out/probe_trace.log 2>&1
tail -3 gpurun_out/probe_trace.log
