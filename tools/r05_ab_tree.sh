#!/bin/bash
# GPU (round 5): the tree / rule kernel changes of this round against the library before them (variants/libczero_base.so):
#   the -m gpu suite on the new library, the sustained search probe A/B (tools/ab_search5.sh), the 1 M-board micro-suite A/B.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -8
bash tools/ab_search5.sh > /dev/null 2>&1
cat gpurun_out/ab_search5.log | python -c "
import sys, json
name = None
for l in sys.stdin:
    l = l.strip()
    if l.startswith('variant='): name = l
    elif l.startswith('{'):
        d = json.loads(l); print(name, d['search_round_ms'], d['nodes'])
    else: print(name, l[:200])
"
: > gpurun_out/ab_micro.log
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export CZ_LIB=$PWD/variants/libczero_base.so; else unset CZ_LIB; fi
    echo "variant=$v rep=$rep" >> gpurun_out/ab_micro.log
    ITERS=10 timeout 200 python tools/micro_rules.py 2>&1 | tail -1 >> gpurun_out/ab_micro.log
  done
done
unset CZ_LIB
cut -c1-400 gpurun_out/ab_micro.log
