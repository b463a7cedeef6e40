#!/bin/bash
# Variants (git-ignored, built beforehand with `python chinesechess-alphazero_amd/build.py --out variants/libczero_<name>.so [-D...]`;
# base / r5a from `git archive <commit>` trees): base = f3d8f07, r5a = 620264f, first2 = -DCZ_FIRST_TERMS=2, tpbscat =
# -DCZ_TPB_SCATTER=1, prof = -DCZ_SIM_PROFILE.
# GPU (round 5, second A/B call): suite on the new default library; search probe base / r5a / new; section profile;
# fused input layer with two table rows per round (variants/libczero_first2.so); k_rules_tpb zero-late scatter variant.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|FAILED" gpurun_out/pytest_gpu.log | tail -8
LOG=gpurun_out/ab_search6.log; : > $LOG
for rep in 1 2; do
  echo "variant=base rep=$rep" >> $LOG
  CZ_LIB=$PWD/variants/libczero_base.so timeout 300 python tools/search_probe.py 2>&1 | tail -1 >> $LOG
  echo "variant=r5a-masks-only rep=$rep" >> $LOG
  CZ_LIB=$PWD/variants/libczero_r5a.so timeout 300 python tools/search_probe.py --masks-only 1 2>&1 | tail -1 >> $LOG
  echo "variant=new-masks-only rep=$rep" >> $LOG
  timeout 300 python tools/search_probe.py --masks-only 1 2>&1 | tail -1 >> $LOG
done
echo "variant=prof-masks-only" >> $LOG
CZ_LIB=$PWD/variants/libczero_prof.so timeout 300 python tools/search_probe.py --masks-only 1 2>&1 | tail -1 >> $LOG
python - <<'PY'
import json
name = None
for l in open("gpurun_out/ab_search6.log"):
    l = l.strip()
    if l.startswith("variant="): name = l
    elif l.startswith("{"):
        d = json.loads(l); print(name, d["search_round_ms"], d.get("cycles_per_sim", ""))
    else: print(name, l[:300])
PY
LOG=gpurun_out/ab_first2.log; : > $LOG
for rep in 1 2; do
  echo "variant=default rep=$rep" >> $LOG
  timeout 200 python tools/time_tower_launches.py c6,c8 32768 masks 2>&1 | grep -v "^{" | tail -2 >> $LOG
  echo "variant=first2 rep=$rep" >> $LOG
  CZ_LIB=$PWD/variants/libczero_first2.so timeout 200 python tools/time_tower_launches.py c6,c8 32768 masks 2>&1 | grep -v "^{" | tail -2 >> $LOG
done
cat $LOG
CZ_LIB=$PWD/variants/libczero_first2.so timeout 600 python -m pytest tests/test_gpu_masks.py tests/test_gpu_c6.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
LOG=gpurun_out/ab_micro2.log; : > $LOG
for rep in 1 2; do
  for v in default tpbscat; do
    if [ $v = tpbscat ]; then export CZ_LIB=$PWD/variants/libczero_tpbscat.so; else unset CZ_LIB; fi
    echo "variant=$v rep=$rep" >> $LOG
    ITERS=10 timeout 200 python tools/micro_rules.py 2>&1 | tail -1 | cut -c100-330 >> $LOG
  done
done
unset CZ_LIB
cat $LOG
CZ_LIB=$PWD/variants/libczero_tpbscat.so timeout 600 python -m pytest tests/test_gpu_rules.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
