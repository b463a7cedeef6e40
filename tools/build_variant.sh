#!/bin/bash
# Build an alternative libczero into variants/libczero_<name>.so (git-ignored, travels with the gpurun snapshot) for
# A/B runs on one GPU box: CZ_LIB=$PWD/variants/libczero_<name>.so selects it (cchess_alphazero/_native.py).
#   bash tools/build_variant.sh prof -DCZ_SIM_PROFILE         # section timers of a simulation (tools/search_probe.py prints them)
#   bash tools/build_variant.sh base                          # the working tree as it is
# then e.g.:  CZ_LIB=$PWD/variants/libczero_prof.so python tools/search_probe.py ; bash tools/ab_search.sh
set -e
name=${1:?usage: build_variant.sh <name> [-DNAME[=V] ...]}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/variants"
python "$ROOT/chinesechess-alphazero_amd/build.py" --out "$ROOT/variants/libczero_$name.so" "$@" -DCZ_VARIANT_$name
echo "built variants/libczero_$name.so"
