#!/usr/bin/env python3
"""Builds libczero.so (hand-written HIP for gfx950) in-tree under csrc/.

    python chinesechess-alphazero_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libczero.so")
SOURCES = ["xq_kernels.hip", "xq_search.hip", "xq_nn_epilogue.hip", "xq_conv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",          # PUCT / backup arithmetic must not be fused (bit-parity with the reference)
         "-fno-fast-math"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "czero.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, probe=False):
    """probe=True adds -DCZ_CONV_PROBE: the tuning variants of tools/conv_probe.py (CZ_CONV_VARIANT) are compiled in."""
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc"] + FLAGS + (["-DCZ_CONV_PROBE"] if probe else []) + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv or "--probe" in sys.argv, probe="--probe" in sys.argv)
