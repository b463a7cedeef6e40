#!/usr/bin/env python3
"""tests/golden/make_golden.py -- generates the golden fixtures in this directory by
IMPORTING THE REFERENCE from /root/reference (read-only).  Run in the build container only;
/root/reference does not exist on the GPU box, so the tests read the committed JSON files.

    python tests/golden/make_golden.py [rules|catch|mcts|games|all]

Fixtures written (NumPy / Python versions recorded inside each file):
  positions_1k.json   -- SURVEY Appendix B.4 suite: state, ordered legal moves,
                         done(need_check=True), planes CRC, per-move step CRC / no_eat bits
  known_answers.json  -- reference test.py inputs (test_done, test_check_and_catch, ...) and B.2 values
  catch_cases.json    -- (state, move) -> will_check_or_catch / be_catched
  mcts_k1.json        -- CChessPlayer visit counts, search_threads=1, stub nets (see stub_net.py)
  games_k1.json       -- full SelfPlayWorker.start_game records (tau=0, noise 0, K=1)
"""
import json
import os
import random
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "cchess_alphazero"))   # for `import configs.mini`
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import cchess_alphazero.environment.static_env as senv  # noqa: E402
from cchess_alphazero.environment.lookup_tables import ActionLabelsRed, flip_move  # noqa: E402


def meta():
    return {"numpy": np.__version__, "python": sys.version.split()[0],
            "generator": "tests/golden/make_golden.py", "reference": "NeymarL/ChineseChess-AlphaZero @ /root/reference"}


def crc(b):
    return zlib.crc32(b) & 0xFFFFFFFF


def position_record(state):
    moves = senv.get_legal_moves(state)
    d = senv.done(state, need_check=True)
    planes = senv.state_to_planes(state)
    steps, no_eat = [], []
    for m in moves:
        try:
            s2, ne = senv.new_step(state, m)
        except ValueError:
            s2, ne = "ValueError", True
        steps.append(s2)
        no_eat.append("1" if ne else "0")
    return {
        "state": state,
        "moves": " ".join(moves),
        "done": [bool(d[0]), int(d[1]), d[2]] + [bool(x) for x in d[3:]],
        "planes_crc": crc(planes.astype(np.float32).tobytes()),
        "planes_sum": int(planes.sum()),
        "step_crc": crc("\n".join(steps).encode()),
        "no_eat": "".join(no_eat),
        "has_attack": bool(senv.has_attack_chessman(state)),
        "flip": senv.fliped_state(state),
    }


SPECIAL = [
    # Appendix B.2 states
    senv.INIT_STATE,
    '4s4/9/4e4/p8/2e2R2p/P5E2/8P/9/9/4S1E2',
    'rkem1cekr/1m7/1c7/p1p3p1p/2p5s/2P1R4/P1P3P1P/1C5C1/9/RKEMSMEK1',
    'rkemsme1r/9/1c3c2k/p1p5p/7p1/3PR4/P1P3P1P/C7C/9/RKEMSMEK1',
    '9/5s3/9/9/2R6/9/7pP/9/5r3/2E1S4',
    '4s1e2/3R5/9/1P7/p8/6r2/9/9/3S5/9',
    '9/5s3/9/9/2R6/8P/6p2/9/5r3/2E1S4',
    'rkemsmekr/9/1c7/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/R8/1KEMSMEcR',
    # kings facing / bare kings / missing kings
    '3s5/9/9/9/9/9/9/9/9/3S5',
    '4s4/9/9/9/9/9/9/9/9/3S5',
    '4s4/9/9/9/4p4/9/9/9/9/4S4',
    '4s4/9/9/9/9/9/9/9/4S4/9',
    '9/9/9/9/9/9/9/9/9/4S4',
    '4s4/9/9/9/9/9/9/9/9/9',
    '3s5/9/9/9/9/9/9/9/3R5/3S5',
    '3s5/9/9/9/9/9/9/9/4R4/3S5',
    # cannon on each edge with 0 / 1 / 2 screens
    'C3s4/9/9/9/9/9/9/9/9/4S4',
    '4s3C/9/9/9/9/9/9/9/9/4S4',
    '4s4/9/9/9/9/9/9/9/9/C3S4',
    '4s4/9/9/9/9/9/9/9/9/4S3C',
    'C1p1s1p2/9/9/9/9/9/9/9/9/4S4',
    'C1p1s1p1r/p8/9/9/r8/9/9/9/9/4S4',
    '4s4/9/9/9/9/9/9/9/9/CP1PS1p1r',
    'r3s4/9/p8/9/9/9/P8/9/9/C3S4',
    'r3s4/9/p8/9/p8/9/9/9/9/C3S4',
    '3cs4/9/9/4c4/9/9/4C4/9/4P4/3CS4',
    # pawns on each side of the river
    '4s4/9/9/9/p1p1p1p1p/P1P1P1P1P/9/9/9/4S4',
    '4s4/9/9/P1P1P1P1P/9/9/p1p1p1p1p/9/9/4S4',
    '4s4/P7P/9/9/9/9/9/9/p7p/4S4',
    'P3s3P/9/9/9/9/9/9/9/9/p3S3p',
    # blocked knights / elephants
    '4s4/9/9/9/9/9/9/1P7/PKP6/1P2S4',
    '4s4/9/9/9/9/2P1P4/3K5/2P1P4/9/4S4',
    '4s4/9/9/9/9/9/9/4p4/3pKp3/4S4',
    '4s4/9/9/9/9/9/1p1p5/2E6/1p1p5/4S4',
    '4s4/9/9/9/9/2E3E2/9/4E4/9/2E1S1E2',
    '2e1s1e2/9/4e4/9/2e3e2/9/9/9/9/4S4',
    # advisors / king in palace corners
    '3s5/9/9/9/9/9/9/3M1M3/4M4/3MSM3',
    '3ms4/4m4/3m1m3/9/9/9/9/9/4S4/9',
    '5s3/9/9/9/9/9/9/5S3/9/9',
    # rook/cannon crowded files
    'r1r1s1r1r/9/9/9/9/9/9/9/9/R1R1S1R1R',
    'c1c1s1c1c/9/9/9/9/9/9/9/9/C1C1S1C1C',
    'rkemsmekr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C2C4/9/RKEMSMEKR',
]


def gen_rules():
    random.seed(20260923)
    seen, order = set(), []
    while len(order) < 960:
        s = senv.INIT_STATE
        for _ in range(150):
            if s not in seen:
                seen.add(s)
                order.append(s)
            if senv.done(s)[0]:
                break
            s = senv.step(s, random.choice(senv.get_legal_moves(s)))
        else:
            if s not in seen:
                seen.add(s)
                order.append(s)
    states = order[:960]
    extra = []
    for s in SPECIAL:
        for t in (s, senv.fliped_state(s)):
            if t not in seen:
                seen.add(t)
                extra.append(t)
    states += extra
    recs = [position_record(s) for s in states]
    out = {"meta": meta(), "n_random": 960, "n_special": len(extra), "positions": recs}
    with open(os.path.join(HERE, "positions_1k.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("positions_1k.json:", len(recs), "positions;",
          sum(1 for r in recs if r["done"][0]), "terminal;",
          sum(1 for r in recs if len(r["done"]) > 3 and r["done"][3]), "in check")

    # known answers: reference test.py inputs + SURVEY Appendix B values, recomputed here
    def perft(state, depth):
        if depth == 0:
            return 1, 0
        if senv.done(state)[0]:
            return 1, 1
        n = t = 0
        for m in senv.get_legal_moves(state):
            a, b = perft(senv.step(state, m), depth - 1)
            n += a
            t += b
        return n, t

    import hashlib
    ka = {
        "meta": meta(),
        "labels_sha256": hashlib.sha256("\n".join(ActionLabelsRed).encode()).hexdigest(),
        "labels_len": len(ActionLabelsRed),
        "labels_2036_2040": ActionLabelsRed[2036:2040],
        "index_0001": ActionLabelsRed.index('0001'), "index_7279": ActionLabelsRed.index('7279'),
        "flip_all_crc": crc(" ".join(flip_move(m) for m in ActionLabelsRed).encode()),
        "init_moves": senv.get_legal_moves(senv.INIT_STATE),
        "init_done": list(senv.done(senv.INIT_STATE, need_check=True)),
        "step_init_0001": senv.step(senv.INIT_STATE, '0001'),
        "perft": [list(perft(senv.INIT_STATE, d)) for d in (1, 2, 3)],
        "test_done": {"state": '4s4/9/4e4/p8/2e2R2p/P5E2/8P/9/9/4S1E2',
                      "done": list(senv.done('4s4/9/4e4/p8/2e2R2p/P5E2/8P/9/9/4S1E2'))},
        "test_check_and_catch": {
            "fen": 'rnba1cbnr/1a7/1c7/p1p3p1p/2p5k/2P1R4/P1P3P1P/1C5C1/9/RNBAKABN1 r',
            "state": senv.fen_to_state('rnba1cbnr/1a7/1c7/p1p3p1p/2p5k/2P1R4/P1P3P1P/1C5C1/9/RNBAKABN1 r'),
            "move": '4454',
            "result": bool(senv.will_check_or_catch(
                senv.fen_to_state('rnba1cbnr/1a7/1c7/p1p3p1p/2p5k/2P1R4/P1P3P1P/1C5C1/9/RNBAKABN1 r'), '4454'))},
        "test_be_catched": {"state": 'rkemsme1r/9/1c3c2k/p1p5p/7p1/3PR4/P1P3P1P/C7C/9/RKEMSMEK1', "move": '4454',
                            "result": bool(senv.be_catched(
                                'rkemsme1r/9/1c3c2k/p1p5p/7p1/3PR4/P1P3P1P/C7C/9/RKEMSMEK1', '4454'))},
        "kings_facing": {"state": '3s5/9/9/9/9/9/9/9/9/3S5',
                         "moves": senv.get_legal_moves('3s5/9/9/9/9/9/9/9/9/3S5'),
                         "done": list(senv.done('3s5/9/9/9/9/9/9/9/9/3S5'))},
        "test_static_env": {"state": senv.step(senv.step(senv.INIT_STATE, '0001'), flip_move('7770'))},
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    print("known_answers.json: perft", ka["perft"])


def gen_catch():
    """(state, move) -> will_check_or_catch / be_catched on playout positions, biased to captures/checks."""
    random.seed(777)
    cases, seen = [], set()
    n_true = 0
    while len(cases) < 600:
        s = senv.INIT_STATE
        for ply in range(120):
            if senv.done(s)[0]:
                break
            moves = senv.get_legal_moves(s)
            if ply >= 6 and random.random() < 0.25:
                # probe up to 3 moves of this position
                for m in random.sample(moves, min(3, len(moves))):
                    if (s, m) in seen:
                        continue
                    seen.add((s, m))
                    w = bool(senv.will_check_or_catch(s, m))
                    b = bool(senv.be_catched(s, m))
                    # keep all positives, thin out double negatives
                    if w or b or random.random() < 0.35:
                        cases.append({"state": s, "move": m, "wcc": w, "bc": b})
                        n_true += w
            s = senv.step(s, random.choice(moves))
    cases = cases[:600]
    with open(os.path.join(HERE, "catch_cases.json"), "w") as f:
        json.dump({"meta": meta(), "cases": cases}, f, separators=(",", ":"))
    print("catch_cases.json:", len(cases), "cases;", sum(c["wcc"] for c in cases), "wcc true;",
          sum(c["bc"] for c in cases), "bc true")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("rules", "all"):
        gen_rules()
    if what in ("catch", "all"):
        gen_catch()
    if what in ("mcts", "all"):
        import make_golden_mcts
        make_golden_mcts.gen_mcts()
    if what in ("games", "all"):
        import make_golden_mcts
        make_golden_mcts.gen_games()
