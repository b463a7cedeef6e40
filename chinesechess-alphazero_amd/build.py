#!/usr/bin/env python3
"""Builds libczero.so (hand-written HIP for gfx950) in-tree under csrc/.

    python chinesechess-alphazero_amd/build.py [--force] [--out PATH] [-DNAME[=V] ...]

Every translation unit is compiled to an object file under csrc/_obj/ (in parallel, only when it or a header changed)
and the objects are linked into csrc/libczero.so.  hipcc cross-compiles without a GPU; the .so is git-ignored but
travels with the gpurun snapshot.  `--out` + `-D...` build a variant library next to the default one (A/B runs:
CZ_LIB=<path> selects it, cchess_alphazero/_native.py); variants get their own object directory.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libczero.so")
SOURCES = ["xq_kernels.hip", "xq_search.hip", "xq_nn_epilogue.hip", "xq_conv.hip", "xq_tower.hip", "xq_heads.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off",          # PUCT / backup arithmetic must not be fused (bit-parity with the reference)
         "-fno-fast-math"]


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "czero.h"))
    return deps


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps() + _sources())


def build(force=False, verbose=True, defines=(), out=None):
    """defines: extra -D flags (variant builds); out: the library to write (default csrc/libczero.so)."""
    lib = out or LIB
    if not force and not defines and not needs_build(lib):
        return lib
    tag = hashlib.sha1(" ".join(sorted(defines)).encode()).hexdigest()[:8] if defines else "default"
    objdir = os.path.join(CSRC, "_obj", tag)
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(d) for d in _deps())

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_t, os.path.getmtime(src)):
            return obj
        cmd = ["hipcc"] + FLAGS + list(defines) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    build(force="--force" in argv, defines=[a for a in argv if a.startswith("-D")], out=out)
