#!/bin/bash
# GPU (round 6, session 1): where the build of round 5's end stands, on ONE box:
#   1. per-launch times of the 7 x 128 tower for every arithmetic the guard can choose  -> gpurun_out/r06_tower_launches.log
#   2. the chain with every block reading the same filters (no L2 refetches; wrong results)  -> gpurun_out/r06_same_filters.log
#   3. FETCH_SIZE / WRITE_SIZE against known byte counts in the copy waves' store patterns   -> gpurun_out/hbm_probe/
set -u
mkdir -p gpurun_out/hbm_probe
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 400 python tools/time_tower_launches.py "c6,c8,c8>3,f16x3" 32768 masks > gpurun_out/r06_tower_launches.log 2>&1
tail -6 gpurun_out/r06_tower_launches.log | cut -c1-400
timeout 300 python tools/time_chain_same_filters.py 32768 8 > gpurun_out/r06_same_filters.log 2>&1
tail -6 gpurun_out/r06_same_filters.log | cut -c1-400
cd /tmp
P=$ROOT/tools/probes/hbm_counter_probe
O=$ROOT/gpurun_out/hbm_probe
timeout 120 $P > $O/probe.json 2> $O/probe.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $P > /dev/null 2> $O/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $P > /dev/null 2> $O/write.err
cd $ROOT
python tools/summarize_hbm_probe.py $O > gpurun_out/r06_hbm_counter_calibration.json 2> gpurun_out/hbm_probe/summary.err
cat gpurun_out/r06_hbm_counter_calibration.json | head -60
find $O -name '*.csv' -size +2M -delete
