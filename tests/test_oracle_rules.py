"""Pins oracle/ (the C restatement) against golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import zlib

import numpy as np

from oracle import xq_oracle as xo


def crc(b):
    return zlib.crc32(b) & 0xFFFFFFFF


def test_label_table(known_answers):
    labels = xo.labels()
    assert len(labels) == known_answers["labels_len"] == 2086
    assert hashlib.sha256("\n".join(labels).encode()).hexdigest() == known_answers["labels_sha256"]
    assert labels[2036:2040] == known_answers["labels_2036_2040"]
    assert labels.index('0001') == known_answers["index_0001"]
    assert labels.index('7279') == known_answers["index_7279"]
    assert crc(" ".join(xo.flip_move(m) for m in labels).encode()) == known_answers["flip_all_crc"]
    fr, to, lo = xo.label_tables()
    assert len(set(zip(fr.tolist(), to.tolist()))) == 2086


def test_known_answers(known_answers):
    ka = known_answers
    assert xo.get_legal_moves(xo.INIT_STATE) == ka["init_moves"]
    assert list(xo.done(xo.INIT_STATE, need_check=True)) == ka["init_done"]
    assert xo.step(xo.INIT_STATE, '0001') == ka["step_init_0001"]
    assert list(xo.done(ka["test_done"]["state"])) == ka["test_done"]["done"]
    c = ka["test_check_and_catch"]
    assert xo.will_check_or_catch(c["state"], c["move"]) == c["result"]
    c = ka["test_be_catched"]
    assert xo.be_catched(c["state"], c["move"]) == c["result"]
    c = ka["kings_facing"]
    assert xo.get_legal_moves(c["state"]) == c["moves"]
    assert list(xo.done(c["state"])) == c["done"]
    assert xo.step(xo.step(xo.INIT_STATE, '0001'), xo.flip_move('7770')) == ka["test_static_env"]["state"]


def test_perft(known_answers):
    def perft(board, depth):
        if depth == 0:
            return 1, 0
        if xo.done_board(board)[0]:
            return 1, 1
        n = t = 0
        for m in xo.legal_moves_board(board):
            a, b = perft(xo.step_board(board, m)[0], depth - 1)
            n += a
            t += b
        return n, t
    b0 = xo.state_to_board(xo.INIT_STATE)
    assert [list(perft(b0, d)) for d in (1, 2, 3)] == known_answers["perft"]


def test_positions_1k(positions_1k):
    assert len(positions_1k) >= 1000
    for r in positions_1k:
        s = r["state"]
        b = xo.state_to_board(s)
        assert xo.board_to_state(b) == s
        moves = xo.get_legal_moves(s)
        assert " ".join(moves) == r["moves"], s
        assert list(xo.done(s, need_check=True)) == r["done"], s
        pl = xo.state_to_planes(s)
        assert crc(pl.tobytes()) == r["planes_crc"] and int(pl.sum()) == r["planes_sum"]
        steps, ne = [], []
        for m in moves:
            try:
                s2, e = xo.new_step(s, m)
            except ValueError:
                s2, e = "ValueError", True
            steps.append(s2)
            ne.append("1" if e else "0")
        assert crc("\n".join(steps).encode()) == r["step_crc"], s
        assert "".join(ne) == r["no_eat"]
        assert xo.has_attack_chessman(s) == r["has_attack"]
        assert xo.fliped_state(s) == r["flip"]


def test_catch_cases(catch_cases):
    assert len(catch_cases) >= 500
    for c in catch_cases:
        assert xo.will_check_or_catch(c["state"], c["move"]) == c["wcc"], c
        assert xo.be_catched(c["state"], c["move"]) == c["bc"], c


def test_batch_matches_scalar(positions_1k):
    boards = np.stack([xo.state_to_board(r["state"]) for r in positions_1k])
    out = xo.batch_rules(boards)
    for i, r in enumerate(positions_1k):
        mv = [xo.label_str(m) for m in out["moves"][i, :out["counts"][i]]]
        assert " ".join(mv) == r["moves"]
        assert bool(out["over"][i]) == r["done"][0] and int(out["v"][i]) == r["done"][1]
