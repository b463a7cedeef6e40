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


def test_quad_movegen_matches_the_oracle(harness, positions_1k):
    """quad_plan / quad_emit (the quad-of-lanes generator of wave_movegen) on the golden suite and on a few thousand
    oracle playout positions, labels from the table and by arithmetic."""
    lab = np.zeros(160, dtype=np.uint16)
    ft = np.zeros(160, dtype=np.uint16)
    boards = [xo.state_to_board(r["state"]) for r in positions_1k]
    rng = np.random.default_rng(11)
    for _ in range(40):
        b = xo.state_to_board(xo.INIT_STATE)
        for _ply in range(120):
            boards.append(b)
            if xo.done_board(b)[0]:
                break
            mv = xo.legal_moves_board(b)
            b, _ = xo.step_board(b, int(mv[rng.integers(len(mv))]))
    for b in boards:
        exp = xo.legal_moves_board(b)
        for gen in (harness.lane_movegen_quad,):
            for formula in (0, 1):
                n = gen(b.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p),
                        ft.ctypes.data_as(C.c_void_p), formula)
                assert n == len(exp) and (lab[:n] == exp).all(), (formula, xo.board_to_state(b))
        n = harness.lane_movegen(b.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), ft.ctypes.data_as(C.c_void_p))
        ft_ref = ft[:n].copy()
        harness.lane_movegen_quad(b.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), ft.ctypes.data_as(C.c_void_p), 1)
        assert (ft[:n] == ft_ref).all()


def test_file_bits_and_quad_on_arbitrary_boards(harness):
    """file_bits against its definition on random square sets; the quad generator against gen_piece on boards no game
    reaches (any piece on any square, up to 40 pieces of the mover)."""
    harness.lane_file_bits_mismatches.argtypes = [C.c_uint64, C.c_int]
    assert harness.lane_file_bits_mismatches(12345, 20000) == 0
    rng = np.random.default_rng(3)
    lab = np.zeros(512, dtype=np.uint16)
    ft = np.zeros(512, dtype=np.uint16)
    lab2 = np.zeros(512, dtype=np.uint16)
    ft2 = np.zeros(512, dtype=np.uint16)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for _ in range(3000):
        b = np.zeros(90, dtype=np.int8)
        k = int(rng.integers(1, 12))
        sq = rng.choice(90, size=k, replace=False)
        b[sq] = rng.integers(-7, 8, size=k).astype(np.int8)
        n = harness.lane_movegen(vp(b), vp(lab), vp(ft))
        if n > 128:
            continue
        for gen in (harness.lane_movegen_quad,):
            for formula in (0, 1):
                n2 = gen(vp(b), vp(lab2), vp(ft2), formula)
                assert n2 == n, (formula, b.tolist())
                assert (ft2[:n] == ft[:n]).all(), (formula, b.tolist())
                if formula == 0:
                    assert (lab2[:n] == lab[:n]).all(), b.tolist()


def test_thread_per_board_rules(harness, positions_1k):
    """xq_tpb.h (one board per GPU lane) run on the CPU: move lists, done(need_check) for the golden suite and a
    few thousand oracle playout positions."""
    lab = np.zeros(160, dtype=np.uint16)
    out = np.zeros(4, dtype=np.int32)
    boards = [xo.state_to_board(r["state"]) for r in positions_1k]
    rng = np.random.default_rng(7)
    for _ in range(40):
        b = xo.state_to_board(xo.INIT_STATE)
        for _ply in range(120):
            boards.append(b)
            if xo.done_board(b)[0]:
                break
            mv = xo.legal_moves_board(b)
            b, _ = xo.step_board(b, int(mv[rng.integers(len(mv))]))
    for b in boards:
        n = harness.tpb_board(b.ctypes.data_as(C.c_void_p), 1, lab.ctypes.data_as(C.c_void_p),
                              out.ctypes.data_as(C.c_void_p))
        exp_moves = xo.legal_moves_board(b)
        assert n == len(exp_moves) and (lab[:n] == exp_moves).all()
        over, v, fm, ck = xo.done_board(b, True)
        assert (bool(out[0]), int(out[1]), int(out[2]), bool(out[3])) == (over, v, fm, ck), xo.board_to_state(b)
