#!/bin/bash
# Build an alternative libczero into variants/libczero_<name>.so (git-ignored, travels with the gpurun snapshot) for
# A/B runs on one GPU box: CZ_LIB=$PWD/variants/libczero_<name>.so selects it (cchess_alphazero/_native.py).
#   bash tools/build_variant.sh quad -DCZ_MOVEGEN_QUAD        # the quad-of-lanes move generator
#   bash tools/build_variant.sh prof -DCZ_SIM_PROFILE         # section timers of a simulation (tools/search_probe.py prints them)
#   bash tools/build_variant.sh base                          # the working tree as it is
# then e.g.:  CZ_LIB=$PWD/variants/libczero_quad.so python -m pytest tests -m gpu -q ; bash tools/ab_search.sh
set -e
name=${1:?usage: build_variant.sh <name> [hipcc flags...]}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/variants"
cd "$ROOT/chinesechess-alphazero_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math "$@" \
    xq_kernels.hip xq_search.hip xq_nn_epilogue.hip xq_conv.hip -o "$ROOT/variants/libczero_$name.so"
echo "built variants/libczero_$name.so"
