"""Runs in its OWN process (tests/test_oracle_live_reference.py starts it): imports the reference from /root/reference
-- whose package name, cchess_alphazero, is also the name of this repository's host package, so the two must not share
an interpreter -- and compares the oracle's restatement with the reference's own functions on freshly drawn positions.

    python tests/live_reference_check.py rules   SEED GAMES MAX_PLIES CAPTURE_BIAS
    python tests/live_reference_check.py history SEED GAMES MAX_PLIES CAPTURE_BIAS
    python tests/live_reference_check.py mcts    SEED N_POSITIONS SIMS
    python tests/live_reference_check.py games   SEED N_GAMES
    python tests/live_reference_check.py arena   SEED N_GAMES
    python tests/live_reference_check.py kgt1    SEED N_POSITIONS K SIMS RUNS

Exit status 0 and a line "ok <mode> <count>" when everything matched; an AssertionError names the first difference.
"""
import os
import random
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)                                   # `from oracle import xq_oracle` (tests/arena_oracle.py)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(REF, "cchess_alphazero"))
sys.path.insert(0, REF)

import cchess_alphazero.environment.static_env as senv  # noqa: E402  (the REFERENCE's)
import xq_oracle as xo  # noqa: E402


def playout_positions(seed, games, max_plies, capture_bias):
    """Random playouts with the reference's own move generator; positions are always 'red to move' strings, as everywhere
    in the reference (step() flips the board)."""
    rng = random.Random(seed)
    out = []
    for _ in range(games):
        state = senv.INIT_STATE
        hist = [state]
        for _ply in range(max_plies):
            moves = senv.get_legal_moves(state)
            if not moves or senv.done(state)[0]:
                break
            out.append((state, list(hist)))
            board = senv.state_to_board(state)
            captures = [m for m in moves if board[int(m[3])][int(m[2])] != '.']
            mv = rng.choice(captures) if captures and rng.random() < capture_bias else rng.choice(moves)
            state = senv.step(state, mv)
            hist.append(mv)
            hist.append(state)
    return out


def check_rules(seed, games, max_plies, capture_bias):
    positions = playout_positions(seed, games, max_plies, capture_bias)
    rng = random.Random(seed + 1)
    for state, _ in positions:
        moves = senv.get_legal_moves(state)
        assert xo.get_legal_moves(state) == moves, state
        want = senv.done(state, need_check=True)
        got = xo.done(state, need_check=True)
        assert (bool(got[0]), int(got[1]), got[2], bool(got[3])) == \
               (bool(want[0]), int(want[1]), want[2], bool(want[3])), state
        assert np.array_equal(np.asarray(xo.state_to_planes(state), dtype=np.float32),
                              senv.state_to_planes(state).astype(np.float32)), state
        assert xo.fliped_state(state) == senv.fliped_state(state)
        assert bool(xo.has_attack_chessman(state)) == bool(senv.has_attack_chessman(state)), state
        for mv in moves:
            try:
                w = senv.new_step(state, mv)
            except ValueError:
                w = None                       # the reference raises when a king is captured (static_env.py:93-95)
            try:
                g = xo.new_step(state, mv)
            except ValueError:
                g = None
            assert (g is None) == (w is None), (state, mv)
            if w is not None:
                assert g[0] == w[0] and bool(g[1]) == bool(w[1]), (state, mv)
        for mv in rng.sample(moves, min(4, len(moves))):
            assert bool(xo.will_check_or_catch(state, mv)) == bool(senv.will_check_or_catch(state, mv)), (state, mv)
            assert bool(xo.be_catched(state, mv)) == bool(senv.be_catched(state, mv)), (state, mv)
    return len(positions)


def check_history(seed, games, max_plies, capture_bias):
    positions = playout_positions(seed, games, max_plies, capture_bias)
    for state, hist in positions:
        want = senv.state_history_to_planes(state, hist)
        got = np.asarray(xo.state_history_to_planes(state, hist), dtype=np.float32)
        assert got.shape == want.shape and np.array_equal(got, want.astype(np.float32)), (state, len(hist))
    return len(positions)


def check_mcts(seed, n_positions, sims):
    """The reference's CChessPlayer (search_threads = 1, no noise, tests/stub_net.py as the network) against the oracle's
    search on positions drawn here: action, visit counts, W (float64 bits) and priors (float32 bits) of every root edge."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_mcts as gm              # configuration / stub helpers of the golden generator (patches sleep)
    import stub_net
    ref_player = gm.ref_player
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    positions = playout_positions(seed, 6, 70, 0.3)
    rng = random.Random(seed)
    picked = rng.sample(positions, n_positions)
    for i, (state, _) in enumerate(picked):
        spec = dict(kind="hash", salt=seed + i)
        c_puct, vl = (1.5, 3) if i % 2 == 0 else (5.0, 1)
        cfg = gm.make_cfg(sims, c_puct=c_puct, vl=vl)
        pipe = stub_net.StubPipe(gm.stub_fn(spec))
        pl = ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False, debugging=False)
        action, _policy = pl.action(state, 0, None)
        want = gm.root_stats(pl, state)
        pl.close()
        ocfg = xo.play_cfg(simulation_num_per_move=sims, search_threads=1, c_puct=c_puct, virtual_loss=vl)
        op = xo.Player(ocfg, spec)
        got_action, _ = op.action(state, 0)
        st = op.node_stats(state)
        op.close()
        assert got_action == action, (state, got_action, action)
        assert " ".join(xo.label_str(m) for m in st["moves"]) == want["moves"], state
        assert [int(x) for x in st["n"]] == want["n"], (state, list(st["n"]), want["n"])
        assert [float(x).hex() for x in st["w"]] == want["w_hex"], state
        assert [float(np.float32(x)).hex() for x in st["p"]] == want["p_hex"], state
    return len(picked)


def check_games(seed, n_games):
    """The reference's own SelfPlayWorker.start_game (tests/golden/make_golden_mcts.py::record_game: stub network, the
    engine's counter-based uniforms behind np.random.choice / random.random) against the oracle's game loop on specs
    drawn here: length, result, every move, the visit-count CRC of every action() call, the evaluation count."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_mcts as gm
    gm._shim_tf()
    import cchess_alphazero.worker.self_play as sp
    rng = random.Random(seed)
    for i in range(n_games):
        spec = dict(name=f"live_{seed}_{i}", salt=seed + i, seed=seed * 7 + i,
                    sims=rng.choice([4, 6, 10, 25, 40]), tau=rng.choice([0.0, 0.9, 0.98]),
                    max_game_length=rng.choice([16, 30, 60]), c_puct=rng.choice([0.5, 1.5, 3.0]))
        if i % 3 == 2:
            spec.update(enable_resign_rate=0.0, resign_threshold=rng.choice([-0.35, 0.3]), min_resign_turn=6)
        gmr = gm.record_game(sp, spec)
        cfg = xo.play_cfg(simulation_num_per_move=spec["sims"], search_threads=1, c_puct=spec["c_puct"],
                          tau_decay_rate=spec["tau"], max_game_length=spec["max_game_length"],
                          enable_resign_rate=spec.get("enable_resign_rate", 1.0),
                          resign_threshold=spec.get("resign_threshold", -0.92),
                          min_resign_turn=spec.get("min_resign_turn", 20))
        r = xo.selfplay_game(cfg, {"kind": "hash", "salt": spec["salt"]}, spec["seed"], 0)
        assert r["turns"] == gmr["turns"], (spec, r["turns"], gmr["turns"])
        assert r["value"] == gmr["value"] and r["store"] == gmr["store"], spec
        if gmr["record"] is not None:
            rec = gmr["record"]
            assert rec[0] == xo.INIT_STATE
            assert r["moves"] == [m for m, _ in rec[1:]], spec
        assert r["visit_crc"][:len(gmr["plies"])].tolist() == [p["crc"] for p in gmr["plies"]], spec
        assert r["counters"]["nn_positions"] == gmr["nn_positions"], spec
    return n_games


def check_arena(seed, n_games):
    """The reference's own EvaluateWorker.start_game (two players, two stub networks, colours by game index, the arena's
    repetition handling; tests/golden/make_golden_mcts.py::_arena_game) against tests/arena_oracle.py over two oracle
    players, on specs drawn here: ply by ply state, action, visit CRC, bans and temperature flags, result."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_mcts as gm
    import stub_net
    from arena_oracle import arena_game
    gm._shim_tf()
    import cchess_alphazero.worker.evaluator as ev
    rng = random.Random(seed)
    endgame = '3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'
    for i in range(n_games):
        spec = dict(name=f"live_arena_{seed}_{i}", idx=i, salts=(seed + 2 * i, seed + 2 * i + 1), seed=seed * 5 + i,
                    sims=rng.choice([4, 6, 8, 12, 30]), max_game_length=rng.choice([20, 40, 80]),
                    c_puct=rng.choice([0.5, 1.0, 1.5]), evaluate=bool(i % 4 == 3))
        if i % 3 == 1:
            spec["init_state"] = endgame
        g = gm._arena_game(ev, spec)
        pc = types.SimpleNamespace(simulation_num_per_move=spec["sims"], search_threads=1, c_puct=spec["c_puct"],
                                   dirichlet_alpha=0.2, tau_decay_rate=0.0, virtual_loss=3,
                                   max_game_length=spec["max_game_length"])
        trace = []
        value, turns, _evals = arena_game(spec["idx"], pc, tuple(dict(kind="hash", salt=x) for x in spec["salts"]),
                                          lambda idx, ply, _s=spec["seed"]: stub_net.philox_uniform(_s, idx, 1, ply),
                                          init_state=spec.get("init_state"), evaluate=spec["evaluate"], trace=trace)
        assert (value, turns) == (g["value"], g["turns"]), (spec, value, turns, g["value"], g["turns"])
        assert len(trace) == len(g["plies"]), spec
        for t, r in zip(trace, g["plies"]):
            assert (t["state"], t["action"], t["crc"], t["sum_n"]) == (r["state"], r["action"], r["crc"], r["sum_n"]), spec
            assert t["no_act"] == r["no_act"] and t["inc"] == r["inc"], spec
    return n_games


def check_kgt1(seed, n_positions, K, sims, runs):
    """search_threads = K > 1: the unmodified reference (its own thread timing: 1 ms sender sleep, 5 ms switch interval,
    as in tests/golden/make_golden_kgt1.py) searched `runs` times per freshly drawn position, against the oracle's
    canonical order.  Where the reference's runs all agree the oracle must give exactly that visit
    vector; where they differ it must have a top move some run has and lie no further from their mean than 1.5 x the
    furthest run (few runs: a crude spread)."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_mcts as gm
    import stub_net
    gm.ref_player.sleep = time.sleep
    sys.setswitchinterval(0.005)
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    positions = playout_positions(seed, 8, 60, 0.3)
    picked = random.Random(seed).sample(positions, n_positions)
    exact_n = deterministic = 0
    for i, (state, _) in enumerate(picked):
        spec = dict(kind="hash", salt=seed + i)
        visits = []
        for _run in range(runs):
            cfg = gm.make_cfg(sims)
            cfg.play.search_threads = K
            pipe = stub_net.StubPipe(gm.stub_fn(spec))
            pl = gm.ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
            pl.action(state, 0, None)
            node = pl.tree[state]
            visits.append([int(node.a[mv].n) if mv in node.a else 0 for mv in node.legal_moves])
            pl.close()
        ocfg = xo.play_cfg(simulation_num_per_move=sims, search_threads=K)
        op = xo.Player(ocfg, spec)
        op.search(state)
        n = [int(x) for x in op.node_stats(state)["n"]]
        op.close()
        assert sum(n) == sum(visits[0]), (state, sum(n), sum(visits[0]))
        v = np.array(visits, dtype=np.float64)
        p = v / v.sum(1, keepdims=True)
        mean = p.mean(0)
        far = float((0.5 * np.abs(p - mean).sum(1)).max())
        q = np.array(n, dtype=np.float64) / sum(n)
        tv = float(0.5 * np.abs(q - mean).sum())
        same = len({tuple(x) for x in visits}) == 1
        exact = tuple(n) in {tuple(x) for x in visits}
        deterministic += same
        exact_n += exact
        print("kgt1", i, "distinct reference vectors", len({tuple(x) for x in visits}), "tv %.4f far %.4f" % (tv, far),
              "exact" if exact else "", flush=True)
        if same:
            assert exact, (state, n, visits[0])
        else:
            assert int(np.argmax(q)) in {int(np.argmax(r)) for r in p}, state
            assert tv <= 1.5 * far + 1e-9, (state, tv, far)
    print("kgt1: deterministic positions", deterministic, "of", n_positions, "; oracle equals a recorded run in", exact_n)
    return n_positions


if __name__ == "__main__":
    mode = sys.argv[1]
    xo.build()
    if mode == "rules":
        n = check_rules(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    elif mode == "history":
        n = check_history(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))
    elif mode == "mcts":
        n = check_mcts(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    elif mode == "games":
        n = check_games(int(sys.argv[2]), int(sys.argv[3]))
    elif mode == "arena":
        n = check_arena(int(sys.argv[2]), int(sys.argv[3]))
    elif mode == "kgt1":
        n = check_kgt1(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
    else:
        raise SystemExit("mode?")
    print("ok", mode, n)
