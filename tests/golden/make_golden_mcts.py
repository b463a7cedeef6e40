"""MCTS / self-play golden vectors, produced by running the REFERENCE's own CChessPlayer and
SelfPlayWorker (imported read-only from /root/reference) against the deterministic stub networks
of tests/stub_net.py.  search_threads = 1 (the only deterministic mode of the reference), noise 0.

Environment control (no reference code is modified):
  * tensorflow / keras are MagicMock modules (absent here; only needed to import worker/self_play.py)
  * np.random.dirichlet -> constant (its value is multiplied by noise_eps = 0)
  * np.random.choice / random.random are replaced by functions that consume the counter-based
    uniform stream (Philox) the engine uses, via NumPy's documented choice algorithm
    (cdf.searchsorted(u, side='right')), so sampled games are reproducible by the engine.
"""
import json
import os
import sys
import zlib
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
for p in (REF, os.path.join(REF, "cchess_alphazero"), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

import stub_net  # noqa: E402  (tests/stub_net.py)

import cchess_alphazero.environment.static_env as senv  # noqa: E402
from cchess_alphazero.config import Config  # noqa: E402
from cchess_alphazero.environment.lookup_tables import ActionLabelsRed  # noqa: E402
import cchess_alphazero.agent.player as ref_player  # noqa: E402

LABEL = {m: i for i, m in enumerate(ActionLabelsRed)}

# The reference's sender thread sleeps 1 ms while HOLDING the queue lock (player.py:113-123), which
# starves the search thread for seconds at a time when the network answers instantly (SURVEY C-12).
# Shorten that sleep and make the interpreter switch threads eagerly: timing only, results unchanged.
import time as _time  # noqa: E402
sys.setswitchinterval(1e-5)
ref_player.sleep = lambda s: _time.sleep(0)


def meta():
    return {"numpy": np.__version__, "python": sys.version.split()[0],
            "generator": "tests/golden/make_golden_mcts.py",
            "reference": "NeymarL/ChineseChess-AlphaZero @ /root/reference, search_threads=1"}


def make_cfg(sims, c_puct=1.5, tau_decay_rate=0.0, vl=3, **kw):
    cfg = Config('mini')
    pc = cfg.play
    pc.simulation_num_per_move = sims
    pc.search_threads = 1
    pc.c_puct = c_puct
    pc.noise_eps = 0
    pc.tau_decay_rate = tau_decay_rate
    pc.virtual_loss = vl
    for k, v in kw.items():
        setattr(pc, k, v)
    return cfg


def stub_fn(spec):
    if spec["kind"] == "uniform":
        return lambda planes: stub_net.uniform_stub_numpy(planes, spec.get("value", 0.0))
    return lambda planes: stub_net.hash_stub_numpy(planes, spec["salt"])


def root_stats(player, state):
    node = player.tree[state]
    moves = node.legal_moves
    n = [int(node.a[m].n) if m in node.a else 0 for m in moves]
    w = [float(node.a[m].w) if m in node.a else 0.0 for m in moves]
    p = [float(np.float32(node.a[m].p)) if m in node.a else 0.0 for m in moves]
    return {"moves": " ".join(moves), "n": n, "w_hex": [float(x).hex() for x in w],
            "p_hex": [float(x).hex() for x in p], "sum_n": int(node.sum_n)}


def visit_crc(moves, n):
    mv = np.array([LABEL[m] for m in moves], dtype=np.uint16)
    nn = np.array(n, dtype=np.int32)
    return zlib.crc32(nn.tobytes(), zlib.crc32(mv.tobytes())) & 0xFFFFFFFF


def gen_mcts():
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    midgame = 'r1e1s1e1r/4m4/2k1c1k2/p1p1p1p1p/9/2P6/P3P1P1P/1CK1C1K2/9/R1EMSME1R'
    endgame = '3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'
    mate1 = '4s4/9/9/9/9/9/9/9/3R5/3S1R3'          # mover can take the king next ply lines nearby
    cases = [
        dict(name="uniform_100", state=senv.INIT_STATE, sims=100, stub=dict(kind="uniform", value=0.0)),
        dict(name="uniform_800", state=senv.INIT_STATE, sims=800, stub=dict(kind="uniform", value=0.0)),
        dict(name="uniform_v025_200", state=senv.INIT_STATE, sims=200, stub=dict(kind="uniform", value=0.25)),
        dict(name="hash1_50", state=senv.INIT_STATE, sims=50, stub=dict(kind="hash", salt=1)),
        dict(name="hash1_800", state=senv.INIT_STATE, sims=800, stub=dict(kind="hash", salt=1)),
        dict(name="hash2_400_c5", state=senv.INIT_STATE, sims=400, c_puct=5.0, stub=dict(kind="hash", salt=2)),
        dict(name="hash3_mid_400", state=midgame, sims=400, stub=dict(kind="hash", salt=3)),
        dict(name="hash4_end_600", state=endgame, sims=600, stub=dict(kind="hash", salt=4)),
        dict(name="hash5_mate_300", state=mate1, sims=300, stub=dict(kind="hash", salt=5)),
        dict(name="hash6_noact_200", state=senv.INIT_STATE, sims=200, stub=dict(kind="hash", salt=6),
             no_act=['1219', '7279', '1242']),
        dict(name="hash7_vl1_300", state=midgame, sims=300, vl=1, stub=dict(kind="hash", salt=7)),
    ]
    out = []
    for c in cases:
        cfg = make_cfg(c["sims"], c_puct=c.get("c_puct", 1.5), vl=c.get("vl", 3))
        pipe = stub_net.StubPipe(stub_fn(c["stub"]))
        pl = ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False, debugging=False)
        action, policy = pl.action(c["state"], 0, c.get("no_act"))
        rec = dict(c)
        rec.update(root_stats(pl, c["state"]))
        rec["action"] = action
        rec["policy_crc"] = zlib.crc32(np.asarray(policy, dtype=np.float64).tobytes()) & 0xFFFFFFFF
        rec["tree_size"] = len(pl.tree)
        rec["nn_positions"] = pipe.n_positions
        out.append(rec)
        pl.close()
        print(c["name"], "action", action, "tree", rec["tree_size"], "evals", pipe.n_positions, flush=True)

    # multi-ply lines with subtree reuse (tau = 0 -> argmax move, deterministic)
    lines = []
    for salt, sims, plies in ((11, 60, 14), (12, 150, 8)):
        cfg = make_cfg(sims)
        pipe = stub_net.StubPipe(stub_fn(dict(kind="hash", salt=salt)))
        pl = ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
        state, steps = senv.INIT_STATE, []
        for turn in range(plies):
            if senv.done(state)[0]:
                break
            before = pipe.n_positions
            action, _ = pl.action(state, turn)
            st = root_stats(pl, state)
            steps.append({"state": state, "action": action, "sum_n": st["sum_n"], "n": st["n"],
                          "moves": st["moves"], "evals": pipe.n_positions - before})
            state = senv.step(state, action)
        pl.close()
        lines.append({"salt": salt, "sims": sims, "steps": steps})
        print("line salt", salt, "plies", len(steps), "evals/ply", [s["evals"] for s in steps], flush=True)

    # 28-plane input (use_history=True, state_history_to_planes): without a game history (self-play), with the
    # `hist` argument of action() (UCI front-end) long enough / too short to reach two plies back
    game = [senv.INIT_STATE]
    for mv in ('7242', '7062', '1219', '0001'):
        game += [mv, senv.step(game[-1], mv)]
    hist_cases = []
    for name, state, hist, salt, sims in (("hist_none", game[-1], None, 51, 200),
                                          ("hist_full", game[-1], list(game), 52, 200),
                                          ("hist_short", game[2], list(game[:3]), 53, 120)):
        cfg = make_cfg(sims)
        pipe = stub_net.StubPipe(stub_fn(dict(kind="hash", salt=salt)))
        pl = ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False, use_history=True)
        action, policy = pl.action(state, 4, None, hist=hist)
        rec = dict(name=name, state=state, hist=hist, sims=sims, stub=dict(kind="hash", salt=salt))
        rec.update(root_stats(pl, state))
        rec["action"] = action
        rec["nn_positions"] = pipe.n_positions
        hist_cases.append(rec)
        pl.close()
        print(name, "action", action, "evals", pipe.n_positions, flush=True)

    with open(os.path.join(HERE, "mcts_k1.json"), "w") as f:
        json.dump({"meta": meta(), "cases": out, "lines": lines, "hist_cases": hist_cases}, f, separators=(",", ":"))


def _mcts_1k_chunk(args):
    """worker process: one K = 1 search per position of the chunk"""
    idx, states, sims, salt = args
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    cfg = make_cfg(sims)
    out = []
    for state in states:
        pipe = stub_net.StubPipe(stub_fn(dict(kind="hash", salt=salt)))
        pl = ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
        action, _ = pl.action(state, 0)
        node = pl.tree[state]
        n = [int(node.a[m].n) if m in node.a else 0 for m in node.legal_moves]
        w = np.array([float(node.a[m].w) if m in node.a else 0.0 for m in node.legal_moves], dtype=np.float64)
        out.append({"crc": visit_crc(node.legal_moves, n), "w_crc": zlib.crc32(w.tobytes()) & 0xFFFFFFFF,
                    "sum_n": int(node.sum_n), "action": action, "evals": pipe.n_positions,
                    "tree_size": len(pl.tree)})
        pl.close()
    print("mcts_1k chunk", idx, "done", flush=True)
    return idx, out


def gen_mcts_1k(sims=800, procs=7):
    """north_star: "visit-count outputs bit-identical to the reference on a fixed 1k-position suite" at "800
    sims/move": one K = 1 search of 800 simulations from every non-terminal position of positions_1k.json that was
    taken from real play (hash stub, salt 101), run by the reference's own CChessPlayer.  ~5 s per search: the
    positions are spread over `procs` worker processes (the searches are independent)."""
    import multiprocessing as mp
    with open(os.path.join(HERE, "positions_1k.json")) as f:
        suite = json.load(f)
    # the 960 positions taken from real (random) play; the hand-made special positions are left out: some of them
    # lead to a node whose mover has no move at all, where the reference's search thread dies and action() hangs
    positions = suite["positions"][:suite["n_random"]]
    salt = 101
    todo = [i for i, r in enumerate(positions) if not (r["done"][0] or not r["moves"])]
    chunks = [todo[i:i + 8] for i in range(0, len(todo), 8)]
    jobs = [(k, [positions[i]["state"] for i in ch], sims, salt) for k, ch in enumerate(chunks)]
    out = [None] * len(positions)          # terminal / no move at all (the reference player dead-locks): null
    with mp.get_context("fork").Pool(procs) as pool:
        for k, res in pool.imap_unordered(_mcts_1k_chunk, jobs):
            for i, r in zip(chunks[k], res):
                out[i] = r
    with open(os.path.join(HERE, "mcts_1k.json"), "w") as f:
        json.dump({"meta": meta(), "sims": sims, "stub": dict(kind="hash", salt=salt), "results": out}, f,
                  separators=(",", ":"))
    print("mcts_1k.json:", sum(1 for x in out if x), "searches of", sims, "simulations")


def _shim_tf():
    for name in ("tensorflow", "keras", "keras.engine", "keras.engine.topology", "keras.engine.training",
                 "keras.layers", "keras.layers.convolutional", "keras.layers.core", "keras.layers.merge",
                 "keras.layers.normalization", "keras.regularizers", "keras.backend", "keras.models",
                 "keras.optimizers", "keras.callbacks", "keras.utils"):
        sys.modules.setdefault(name, MagicMock())


def record_game(sp, s):
    """One game of the reference's own SelfPlayWorker.start_game under the environment control described at the top
    (stub network, Philox-driven np.random.choice / random.random, constant Dirichlet); s: a spec dict as in
    gen_games.  sp: the imported cchess_alphazero.worker.self_play module."""
    from collections import defaultdict
    cfg = make_cfg(s["sims"], c_puct=s.get("c_puct", 1.5), tau_decay_rate=s["tau"],
                   max_game_length=s["max_game_length"],
                   enable_resign_rate=s.get("enable_resign_rate", 1.0),
                   resign_threshold=s.get("resign_threshold", -0.92),
                   min_resign_turn=s.get("min_resign_turn", 20))
    cfg.play_data.nb_game_in_file = 1
    seed, game_id = s["seed"], 0
    calls = {"choice": 0, "random": 0}
    ply_log = []

    def fake_choice(a, p=None, _c=calls):
        u = stub_net.philox_uniform(seed, game_id, 1, _c["choice"])
        _c["choice"] += 1
        return stub_net.numpy_choice(p, u)

    def fake_random(_c=calls):
        u = stub_net.philox_uniform(seed, game_id, 0, _c["random"])
        _c["random"] += 1
        return u

    np.random.choice = fake_choice
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    sp.random = fake_random

    orig_action = ref_player.CChessPlayer.action

    def logged_action(self, state, turns, no_act=None, depth=None, infinite=False, hist=None,
                      increase_temp=False, _orig=orig_action, _log=ply_log):
        r = _orig(self, state, turns, no_act, depth, infinite, hist, increase_temp)
        node = self.tree[state]
        n = [int(node.a[m].n) if m in node.a else 0 for m in node.legal_moves]
        _log.append({"crc": visit_crc(node.legal_moves, n), "sum_n": int(node.sum_n),
                     "no_act": list(no_act or []), "inc": bool(increase_temp)})
        return r

    ref_player.CChessPlayer.action = logged_action
    sp.CChessPlayer.action = logged_action
    saved = {}

    def fake_save(self, idx, data, _s=saved):
        _s["data"] = data

    sp.SelfPlayWorker.save_play_data = fake_save
    sp.SelfPlayWorker.remove_play_data = lambda self: None
    pipe = stub_net.StubPipe(stub_fn(dict(kind="hash", salt=s["salt"])))
    worker = sp.SelfPlayWorker(cfg, pipes=[pipe], pid=0, use_history=False)
    v, turns, state, store = worker.start_game(1, defaultdict(ref_player.VisitState))
    ref_player.CChessPlayer.action = orig_action
    sp.CChessPlayer.action = orig_action
    rec = dict(s)
    rec.update({"value": v, "turns": turns, "final_state": state, "store": bool(store),
                "record": saved.get("data"), "plies": ply_log, "nn_positions": pipe.n_positions,
                "n_random_calls": calls["random"], "n_choice_calls": calls["choice"]})
    print(s["name"], "turns", turns, "value", v, "store", store, "evals", pipe.n_positions,
          "no_act plies", sum(1 for p in ply_log if p["no_act"]),
          "inc_temp plies", sum(1 for p in ply_log if p["inc"]), flush=True)
    return rec


def gen_games():
    _shim_tf()
    import cchess_alphazero.worker.self_play as sp
    from collections import defaultdict

    specs = [
        dict(name="argmax_a", salt=21, sims=40, tau=0.0, max_game_length=30, seed=1001),
        dict(name="argmax_b", salt=22, sims=25, tau=0.0, max_game_length=40, seed=1002),
        dict(name="sampled_a", salt=23, sims=40, tau=0.98, max_game_length=25, seed=1003),
        dict(name="sampled_b", salt=24, sims=30, tau=0.9, max_game_length=30, seed=1004),
        dict(name="resign", salt=25, sims=40, tau=0.0, max_game_length=40, seed=1005,
             enable_resign_rate=0.0, resign_threshold=-0.35, min_resign_turn=6),
        dict(name="short_c3", salt=26, sims=30, tau=0.98, max_game_length=12, seed=1006, c_puct=3.0),
        # near-random play (tiny searches): games that end by king capture (final_move) and early
        dict(name="blunder_a", salt=27, sims=5, tau=0.98, max_game_length=60, seed=1007),
        dict(name="blunder_b", salt=28, sims=6, tau=0.98, max_game_length=60, seed=1008),
        dict(name="blunder_c", salt=29, sims=8, tau=0.9, max_game_length=60, seed=1009),
        dict(name="blunder_d", salt=30, sims=4, tau=0.98, max_game_length=60, seed=1010),
        # resignation: best root Q below a high threshold
        dict(name="resign_hi", salt=31, sims=30, tau=0.98, max_game_length=40, seed=1011,
             enable_resign_rate=0.0, resign_threshold=0.3, min_resign_turn=6),
        # deterministic shallow play: position repetitions -> no_act / increase_temp / idle-loop draw
        dict(name="repeat_a", salt=32, sims=8, tau=0.0, max_game_length=60, seed=1012),
        dict(name="repeat_b", salt=33, sims=12, tau=0.0, max_game_length=60, seed=1013),
        dict(name="repeat_c", salt=34, sims=10, tau=0.0, max_game_length=60, seed=1014, c_puct=0.5),
        # the production search size: 800 simulations per move with subtree reuse over a whole (short) game
        dict(name="prod_800", salt=35, sims=800, tau=0.9, max_game_length=12, seed=1015),
    ]
    only = os.environ.get("GOLDEN_ONLY")          # regenerate just these games and merge them into the file
    old = {}
    if only:
        with open(os.path.join(HERE, "games_k1.json")) as f:
            old = {g["name"]: g for g in json.load(f)["games"]}
        specs = [sp for sp in specs if sp["name"] in only.split(",") or sp["name"] not in old]
    games = [record_game(sp, s) for s in specs]
    if only:
        new = {g["name"]: g for g in games}
        games = [new.get(n, g) for n, g in old.items()] + [g for n, g in new.items() if n not in old]
    with open(os.path.join(HERE, "games_k1.json"), "w") as f:
        json.dump({"meta": meta(), "games": games}, f, separators=(",", ":"))


def _arena_game(ev, s):
    cfg = make_cfg(s["sims"], c_puct=s.get("c_puct", 1.0), tau_decay_rate=0.0,
                   max_game_length=s["max_game_length"])
    cfg.opts.evaluate = bool(s.get("evaluate", False))
    seed, idx = s["seed"], s["idx"]
    calls = {"choice": 0}
    ply_log = []

    def fake_choice(a, p=None, _c=calls):
        u = stub_net.philox_uniform(seed, idx, 1, _c["choice"])
        _c["choice"] += 1
        return stub_net.numpy_choice(p, u)

    class Playouts:                     # randint(8, 12) * 100 -> the simulations of this golden game
        def __mul__(self, other, _n=s["sims"]):
            return _n

    np.random.choice = fake_choice
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    ev.randint = lambda a, b: Playouts()
    orig_action = ref_player.CChessPlayer.action

    def logged_action(self, state, turns, no_act=None, depth=None, infinite=False, hist=None,
                      increase_temp=False, _orig=orig_action, _log=ply_log):
        r = _orig(self, state, turns, no_act, depth, infinite, hist, increase_temp)
        node = self.tree[state]
        n = [int(node.a[m].n) if m in node.a else 0 for m in node.legal_moves]
        _log.append({"state": state, "action": r[0], "crc": visit_crc(node.legal_moves, n),
                     "sum_n": int(node.sum_n), "no_act": None if no_act is None else list(no_act),
                     "inc": bool(increase_temp), "tree": len(self.tree)})
        return r

    ref_player.CChessPlayer.action = logged_action
    ev.CChessPlayer.action = logged_action
    pipes = [stub_net.StubPipe(stub_fn(dict(kind="hash", salt=x))) for x in s["salts"]]
    worker = ev.EvaluateWorker(cfg, [pipes[0]], [pipes[1]], pid=0)
    # start_game always begins at senv.INIT_STATE (:170); endgame starts (where positions repeat and perpetual
    # checks / chases occur within a short game) are fed in through that module attribute
    init_saved = ev.senv.INIT_STATE
    ev.senv.INIT_STATE = s.get("init_state", init_saved)
    try:
        value, turns = worker.start_game(idx)
    finally:
        ev.senv.INIT_STATE = init_saved
        ref_player.CChessPlayer.action = orig_action
        ev.CChessPlayer.action = orig_action
    rec = dict(s)
    rec.update({"value": value, "turns": turns, "plies": ply_log,
                "nn_positions": [pp.n_positions for pp in pipes], "n_choice_calls": calls["choice"]})
    print(s["name"], "idx", idx, "turns", turns, "value", value, "evals", rec["nn_positions"],
          "no_act plies", sum(1 for p in ply_log if p["no_act"]),
          "inc plies", sum(1 for p in ply_log if p["inc"]), flush=True)
    return rec


def gen_arena():
    """EvaluateWorker.start_game (worker/evaluator.py:147-250) recorded from the reference itself: two players with
    their own trees and their own (stub) networks, colours by game index, the arena's own repetition handling
    (before the move, no be_catched branch).  randint (the playout lottery, :153) is pinned per game; the move
    sampling consumes the Philox stream (seed, idx, stream 1, ply) like the engine's arena."""
    _shim_tf()
    import cchess_alphazero.worker.evaluator as ev

    specs = [
        dict(name="even_a", idx=0, salts=(41, 42), sims=30, max_game_length=30, seed=2001),
        dict(name="odd_a", idx=1, salts=(41, 42), sims=30, max_game_length=30, seed=2001),
        dict(name="even_b", idx=2, salts=(43, 44), sims=50, max_game_length=20, seed=2002, c_puct=1.5),
        dict(name="odd_b", idx=3, salts=(43, 44), sims=50, max_game_length=20, seed=2002, c_puct=1.5),
        # shallow deterministic play: positions repeat -> no_act / increase_temp / idle-loop draw before the move
        dict(name="repeat_a", idx=4, salts=(45, 46), sims=8, max_game_length=60, seed=2003),
        dict(name="repeat_b", idx=5, salts=(47, 48), sims=10, max_game_length=60, seed=2004),
        dict(name="repeat_c", idx=6, salts=(49, 50), sims=12, max_game_length=60, seed=2005, c_puct=0.5),
        dict(name="repeat_d", idx=7, salts=(51, 52), sims=6, max_game_length=80, seed=2006),
        # near-random play: king captures (final_move), early endings
        dict(name="blunder_a", idx=8, salts=(53, 54), sims=4, max_game_length=80, seed=2007),
        dict(name="blunder_b", idx=9, salts=(55, 56), sims=5, max_game_length=80, seed=2008),
        # endgame starts: repeated positions before the move -> increase_temp (tau 0.5 sampling), the idle-loop draw
        # after three free repetitions, and a banned move (no_act) from a perpetual check
        dict(name="end_inc", idx=1, salts=(63, 163), sims=8, max_game_length=40, seed=3063,
             init_state='3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'),
        dict(name="end_idle_draw", idx=1, salts=(63, 163), sims=16, max_game_length=40, seed=3063,
             init_state='3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'),
        dict(name="end_no_act", idx=1, salts=(65, 165), sims=16, max_game_length=40, seed=3065,
             init_state='4s4/9/9/9/9/9/9/9/3R5/3S1R3'),
        dict(name="end_inc_even", idx=0, salts=(66, 166), sims=8, max_game_length=40, seed=3066,
             init_state='3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'),
        # config.opts.evaluate = True (compute_elo.py:88): argmax even on repeated positions
        dict(name="elo_end_inc", idx=1, salts=(63, 163), sims=8, max_game_length=40, seed=3063, evaluate=True,
             init_state='3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'),
        dict(name="elo_odd", idx=11, salts=(57, 58), sims=20, max_game_length=40, seed=2010, evaluate=True),
    ]
    games = [_arena_game(ev, s) for s in specs]
    with open(os.path.join(HERE, "arena_k1.json"), "w") as f:
        json.dump({"meta": meta(), "games": games}, f, separators=(",", ":"))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("mcts", "all"):
        gen_mcts()
    if what in ("games", "all"):
        gen_games()
    if what in ("arena", "all"):
        gen_arena()
    if what in ("mcts1k", "all"):
        gen_mcts_1k(sims=int(sys.argv[2]) if len(sys.argv) > 2 else 800)
