"""Pins oracle/xq_mcts.c (the C restatement of agent/player.py and worker/self_play.py) against the
golden vectors recorded from the reference's own CChessPlayer / SelfPlayWorker (search_threads=1).  CPU only."""
import json
import os
import zlib

import numpy as np
import pytest

import stub_net
from oracle import xq_oracle as xo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    with open(path) as f:
        return json.load(f)


def test_stub_nets_agree():
    boards = [xo.state_to_board(xo.INIT_STATE), xo.state_to_board('3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4')]
    planes = np.stack([xo.planes_board(b) for b in boards])
    p, v = stub_net.hash_stub_numpy(planes, 9)
    import torch
    pt, vt = stub_net.hash_stub_torch(torch.from_numpy(planes), 9)
    assert np.array_equal(pt.numpy(), p) and np.array_equal(vt.numpy(), v)
    # the C stub, through a 1-simulation search of each position: the root priors are p / sum(p) in float32
    for i, b in enumerate(boards):
        pl = xo.Player(xo.play_cfg(simulation_num_per_move=2), {"kind": "hash", "salt": 9})
        pl.search(b)
        st = pl.node_stats(b)
        raw = p[i][st["moves"]]
        s = np.float32(0)
        for x in raw:
            s = np.float32(s + x)
        assert np.array_equal(st["p"], (raw / s).astype(np.float32))
        pl.close()
    assert xo.philox_uniform(1003, 5, 1, 7) == stub_net.philox_uniform(1003, 5, 1, 7)


def test_reference_searches():
    data = _golden("mcts_k1.json")
    for c in data["cases"]:
        cfg = xo.play_cfg(simulation_num_per_move=c["sims"], search_threads=1, c_puct=c.get("c_puct", 1.5),
                          virtual_loss=c.get("vl", 3))
        pl = xo.Player(cfg, c["stub"])
        a, pol = pl.action(c["state"], 0, c.get("no_act"), False, 0.5)
        st = pl.node_stats(c["state"])
        assert " ".join(xo.label_str(int(m)) for m in st["moves"]) == c["moves"], c["name"]
        assert st["n"].tolist() == c["n"], c["name"]
        assert st["sum_n"] == c["sum_n"]
        assert [float(x).hex() for x in st["w"]] == c["w_hex"], c["name"]
        assert [float(x).hex() for x in st["p"]] == c["p_hex"], c["name"]
        assert a == c["action"]
        assert zlib.crc32(pol.tobytes()) & 0xFFFFFFFF == c["policy_crc"], c["name"]
        assert pl.tree_size() == c["tree_size"] and pl.counters()["nn_positions"] == c["nn_positions"]
        pl.close()


def test_reference_history_planes():
    """use_history=True (28 input planes) incl. the action(hist=...) quirk of player.py:217-218."""
    data = _golden("mcts_k1.json")
    assert len(data.get("hist_cases", [])) >= 3
    for c in data["hist_cases"]:
        cfg = xo.play_cfg(simulation_num_per_move=c["sims"], search_threads=1, use_history=1)
        pl = xo.Player(cfg, c["stub"])
        pl.set_history(c["hist"])
        a, _ = pl.action(c["state"], 4, None, False, 0.5)
        st = pl.node_stats(c["state"])
        assert st["n"].tolist() == c["n"] and st["sum_n"] == c["sum_n"], c["name"]
        assert [float(x).hex() for x in st["w"]] == c["w_hex"], c["name"]
        assert a == c["action"] and pl.counters()["nn_positions"] == c["nn_positions"]
        pl.close()


def _visit_crc(moves, n):
    return zlib.crc32(np.asarray(n, dtype=np.int32).tobytes(),
                      zlib.crc32(np.asarray(moves, dtype=np.uint16).tobytes())) & 0xFFFFFFFF


def test_reference_visit_counts_on_the_1k_suite(positions_1k):
    """north_star: visit counts identical to the reference on the fixed 1k-position suite (one 48-simulation K = 1
    search from every non-terminal position, recorded from the reference player)."""
    data = _golden("mcts_1k.json")
    res = data["results"]
    assert len(res) == 960 and sum(1 for r in res if r) > 900      # the real-play part of the suite
    cfg = xo.play_cfg(simulation_num_per_move=data["sims"], search_threads=1)
    for r, pos in zip(res, positions_1k):
        if r is None:
            assert pos["done"][0] or not pos["moves"]
            continue
        pl = xo.Player(cfg, data["stub"])
        a, _ = pl.action(pos["state"], 0, None, False, 0.5)
        st = pl.node_stats(pos["state"])
        assert _visit_crc(st["moves"], st["n"]) == r["crc"], pos["state"]
        assert zlib.crc32(st["w"].tobytes()) & 0xFFFFFFFF == r["w_crc"], pos["state"]
        assert st["sum_n"] == r["sum_n"] and a == r["action"] and pl.counters()["nn_positions"] == r["evals"]
        pl.close()


def test_reference_lines_with_subtree_reuse():
    data = _golden("mcts_k1.json")
    for line in data["lines"]:
        cfg = xo.play_cfg(simulation_num_per_move=line["sims"], search_threads=1)
        pl = xo.Player(cfg, {"kind": "hash", "salt": line["salt"]})
        prev = 0
        for turn, step in enumerate(line["steps"]):
            a, _ = pl.action(step["state"], turn, None, False, 0.5)
            st = pl.node_stats(step["state"])
            assert st["n"].tolist() == step["n"] and st["sum_n"] == step["sum_n"], (line["salt"], turn)
            assert a == step["action"]
            ev = pl.counters()["nn_positions"]
            assert ev - prev == step["evals"]
            prev = ev
        pl.close()


def test_reference_games():
    data = _golden("games_k1.json")
    for gm in data["games"]:
        cfg = xo.play_cfg(simulation_num_per_move=gm["sims"], search_threads=1, c_puct=gm.get("c_puct", 1.5),
                          tau_decay_rate=gm["tau"], max_game_length=gm["max_game_length"],
                          enable_resign_rate=gm.get("enable_resign_rate", 1.0),
                          resign_threshold=gm.get("resign_threshold", -0.92),
                          min_resign_turn=gm.get("min_resign_turn", 20))
        r = xo.selfplay_game(cfg, {"kind": "hash", "salt": gm["salt"]}, gm["seed"], 0)
        assert r["turns"] == gm["turns"], gm["name"]
        assert r["value"] == gm["value"] and r["store"] == gm["store"], gm["name"]
        if gm["record"] is not None:
            rec = gm["record"]
            assert rec[0] == xo.INIT_STATE
            assert r["moves"] == [m for m, _ in rec[1:]], gm["name"]
            v = gm["value"]
            assert [x for _, x in rec[1:]] == [v * (-1) ** i for i in range(gm["turns"])]
        assert r["visit_crc"][:len(gm["plies"])].tolist() == [p["crc"] for p in gm["plies"]], gm["name"]
        assert r["counters"]["nn_positions"] == gm["nn_positions"]


def test_reference_arena_games():
    """the arena loop over two oracle players (tests/arena_oracle.py) reproduces the games recorded
    from the reference's own EvaluateWorker.start_game (two trees, colours by game index, repetition handling before
    the move, tau = 0.5 on repeated positions unless config.opts.evaluate, king-capture endings, idle-loop draws)."""
    import types
    from arena_oracle import arena_game
    data = _golden("arena_k1.json")
    assert len(data["games"]) >= 8
    assert any(any(p["no_act"] for p in g["plies"]) for g in data["games"])          # a banned move
    assert any(g["turns"] > len(g["plies"]) for g in data["games"])                  # a king capture (final_move)
    for gm in data["games"]:
        pc = types.SimpleNamespace(simulation_num_per_move=gm["sims"], search_threads=1, c_puct=gm.get("c_puct", 1.0),
                                   dirichlet_alpha=0.2, tau_decay_rate=0.0, virtual_loss=3,
                                   max_game_length=gm["max_game_length"])
        specs = tuple(dict(kind="hash", salt=x) for x in gm["salts"])
        trace = []
        value, turns, evals = arena_game(gm["idx"], pc, specs,
                                         lambda idx, ply, _s=gm["seed"]: stub_net.philox_uniform(_s, idx, 1, ply),
                                         init_state=gm.get("init_state"), evaluate=gm.get("evaluate", False),
                                         trace=trace)
        assert (value, turns) == (gm["value"], gm["turns"]), gm["name"]
        assert len(trace) == len(gm["plies"]), gm["name"]
        for t, r in zip(trace, gm["plies"]):
            assert (t["state"], t["action"], t["crc"], t["sum_n"]) == (r["state"], r["action"], r["crc"], r["sum_n"]), gm["name"]
            assert t["no_act"] == r["no_act"] and t["inc"] == r["inc"], gm["name"]
        assert evals == gm["nn_positions"], gm["name"]


def spread_verdict(case, n):
    """Is the visit vector `n` (root edges in move order) a plausible member of the reference's own population for this
    case?  Returns (total-variation distance to the mean of the reference runs, the furthest reference run's distance,
    whether some reference run has the same top move, whether `n` is exactly one of the recorded vectors)."""
    v = np.array(case["visits"], dtype=np.float64)
    p = v / v.sum(1, keepdims=True)
    mean = p.mean(0)
    spread = 0.5 * np.abs(p - mean).sum(1)
    q = np.asarray(n, dtype=np.float64)
    q = q / q.sum()
    tv = 0.5 * np.abs(q - mean).sum()
    top_ok = int(np.argmax(q)) in {int(np.argmax(r)) for r in p}
    exact = tuple(int(x) for x in n) in {tuple(int(y) for y in x) for x in case["visits"]}
    return float(tv), float(spread.max()), top_ok, exact


def check_against_spread(cases, search_fn):
    """The K > 1 criterion, shared by the CPU oracle test below and the HIP test (tests/test_gpu_search.py): for every
    recorded case the searched visit vector n (search_fn(case) -> (n in move order, sum_n)) must have the reference's
    totals, leave banned moves unvisited, have a top move some reference run has and lie no further from the mean of
    the reference runs than the furthest reference run does (total-variation distance, 1e-9 slack for the cases where
    every reference run gave the same vector).  Beyond plausibility: where the reference is deterministic (all 32 runs
    equal) the result must BE that vector, and over all cases it must be exactly one of the recorded reference vectors
    in at least 20 of 24."""
    exact_n = 0
    for c in cases:
        n, sum_n, moves = search_fn(c)
        assert sum_n == c["sims"] == c["sum_n"][0], c["name"]
        assert int(np.sum(n)) == int(sum(c["visits"][0])), c["name"]
        for mv in c.get("no_act") or []:
            assert n[list(moves).index(xo.label_of_str(mv))] == 0, c["name"]
        tv, far, top_ok, exact = spread_verdict(c, n)
        assert tv <= far + 1e-9, (c["name"], tv, far)
        assert top_ok, c["name"]
        if len({tuple(v) for v in c["visits"]}) == 1:
            assert exact, c["name"]
        exact_n += exact
    assert exact_n >= 20, exact_n
    return exact_n


def test_canonical_order_lies_inside_the_reference_spread():
    """search_threads > 1 in the reference is a thread race, so K > 1 parity is defined against the canonical order of
    DESIGN.md section 3.  tests/golden/kgt1_spread.json holds what the UNMODIFIED reference does (its own 1 ms sender
    sleep, CPython's 5 ms switch interval): 32 runs of the same 800-simulation search at K = 8 and at K = 40 for each of
    12 positions -- opening, middlegames, endgames, a mating position where the proven-win shortcut fires, a search
    under a ban list.  At K = 8 the reference is nearly deterministic (1-7 distinct visit vectors in 32 runs), at K = 40
    it is not (19-32).  The canonical order must reproduce it: inside the spread everywhere, EQUAL to the reference
    where the reference is deterministic, and exactly one of the recorded vectors in >= 20 of the 24 cases (it is in
    23).  The same check runs on the HIP engine: tests/test_gpu_search.py::test_hip_search_lies_inside_the_reference_spread."""
    data = _golden("kgt1_spread.json")
    assert len(data["cases"]) >= 24 and {c["K"] for c in data["cases"]} == {8, 40}
    assert all(len(c["visits"]) >= 32 and c["sims"] == 800 for c in data["cases"])

    def search(c):
        cfg = xo.play_cfg(simulation_num_per_move=c["sims"], search_threads=c["K"])
        pl = xo.Player(cfg, {"kind": "hash", "salt": c["salt"]})
        pl.search(c["state"], 0, c.get("no_act"))
        st = pl.node_stats(c["state"])
        pl.close()
        return st["n"], st["sum_n"], st["moves"]
    assert check_against_spread(data["cases"], search) >= 23


def test_sampling_matches_numpy_choice():
    rng = np.random.default_rng(5)
    cfg = xo.play_cfg(tau_decay_rate=0.98)
    for trial in range(300):
        counts = np.zeros(2086)
        idx = rng.choice(2086, size=rng.integers(2, 45), replace=False)
        counts[idx] = rng.integers(0, 200, size=len(idx))
        if counts.sum() == 0:
            continue
        policy = counts / counts.sum()
        turns = int(rng.integers(0, 40))
        inc = bool(rng.integers(0, 2))
        u = float(rng.random())
        tau = 0.98 ** (turns + 1) if turns < 30 else 0
        if tau < 0.1:
            tau = 0
        if inc:
            tau = 0.5
        if tau == 0:
            exp = int(np.argmax(policy))
        else:
            ret = np.power(policy, 1 / tau)
            ret /= np.sum(ret)
            exp = stub_net.numpy_choice(ret, u)
        assert xo.sample_action(cfg, policy, turns, inc, u) == exp


def test_k_gt_1_is_deterministic_and_complete():
    cfg = xo.play_cfg(simulation_num_per_move=203, search_threads=8)
    a = xo.Player(cfg, {"kind": "hash", "salt": 2})
    b = xo.Player(cfg, {"kind": "hash", "salt": 2})
    a.search(xo.INIT_STATE)
    b.search(xo.INIT_STATE)
    sa, sb = a.node_stats(xo.INIT_STATE), b.node_stats(xo.INIT_STATE)
    assert np.array_equal(sa["n"], sb["n"]) and np.array_equal(sa["w"], sb["w"])
    assert sa["sum_n"] == 203 and sa["n"].sum() == 202          # root expansion takes one simulation
    c = a.counters()
    assert c["sims"] == 203 and c["parked"] > 0
    a.close(); b.close()
