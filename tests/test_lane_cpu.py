"""CPU check of the HIP engine's per-lane rule functions (csrc/xq_lane.h, compiled with g++ into
a throw-away harness) against the oracle on the golden suite.  Catches rule/ordering/table
mistakes before a GPU run; the wave-level glue is covered by the -m gpu tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import xq_oracle as xo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("lane") / "liblane.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "lane_harness.cpp"), "-o", str(out)])
    return C.CDLL(str(out))


def test_tables_match_oracle(harness):
    lo = np.zeros(90 * 90, dtype=np.uint16)
    ft = np.zeros(2086, dtype=np.uint16)
    harness.lane_tables(lo.ctypes.data_as(C.c_void_p), ft.ctypes.data_as(C.c_void_p))
    fr, to, lab = xo.label_tables()
    assert (lo.reshape(90, 90) == lab).all()
    assert ((ft >> 8) == fr).all() and ((ft & 0xFF) == to).all()
    assert harness.lane_nibble_roundtrip() == 1
    assert harness.lane_label_formula_mismatches() == 0


def test_movegen_and_planes(harness, positions_1k):
    lab = np.zeros(128, dtype=np.uint16)
    ft = np.zeros(128, dtype=np.uint16)
    pl = np.zeros(1260, dtype=np.float32)
    for r in positions_1k:
        b = xo.state_to_board(r["state"])
        n = harness.lane_movegen(b.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p),
                                 ft.ctypes.data_as(C.c_void_p))
        assert " ".join(xo.label_str(m) for m in lab[:n]) == r["moves"], r["state"]
        harness.lane_planes(b.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(C.c_void_p))
        assert (pl.reshape(14, 10, 9) == xo.planes_board(b)).all()
