"""-m gpu: the reference-shaped entry points on top of the engine -- CChessPlayer (pipe protocol of
agent/api.py), SelfPlayWorker + play-record files (worker/self_play.py, lib/data_helper.py)."""
import json
import os
import zlib

import numpy as np
import pytest

import stub_net
from oracle import xq_oracle as xo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg(tmp_path, monkeypatch, **play):
    monkeypatch.setenv("DATA_DIR", str(tmp_path / "data"))
    monkeypatch.setenv("PROJECT_DIR", str(tmp_path))
    from cchess_alphazero.config import Config
    cfg = Config('mini')
    for k, v in play.items():
        setattr(cfg.play, k, v)
    return cfg


def test_player_facade_reproduces_reference_player(tmp_path, monkeypatch):
    """CChessPlayer.action through the reference's pipe protocol (send / poll / recv) against the
    visit distribution recorded from the reference's own player (tests/golden/mcts_k1.json)."""
    from cchess_alphazero.agent.player import CChessPlayer
    with open(os.path.join(GOLDEN, "mcts_k1.json")) as f:
        cases = {c["name"]: c for c in json.load(f)["cases"]}
    for name in ("hash1_50", "hash6_noact_200", "uniform_100"):
        c = cases[name]
        cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=c["sims"], search_threads=1, noise_eps=0,
                   tau_decay_rate=0, c_puct=c.get("c_puct", 1.5))
        spec = c["stub"]
        fn = (lambda p, s=spec: stub_net.uniform_stub_numpy(p, s.get("value", 0.0))) if spec["kind"] == "uniform" \
            else (lambda p, s=spec: stub_net.hash_stub_numpy(p, s["salt"]))
        pipe = stub_net.StubPipe(fn)
        pl = CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
        action, policy = pl.action(c["state"], 0, c.get("no_act"))
        assert action == c["action"], name
        assert zlib.crc32(np.asarray(policy, dtype=np.float64).tobytes()) & 0xFFFFFFFF == c["policy_crc"], name
        node = pl.tree[c["state"]]
        assert [node.a[m].n for m in node.legal_moves] == c["n"] and node.sum_n == c["sum_n"]
        assert pipe.n_positions == c["nn_positions"]
        pl.close()


def test_selfplay_worker_writes_reference_records(tmp_path, monkeypatch):
    import torch
    from cchess_alphazero.agent.model import CChessModel
    from cchess_alphazero.lib.data_helper import get_game_data_filenames, read_game_data_from_file
    from cchess_alphazero.worker.self_play import SelfPlayWorker
    cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=12, search_threads=4, max_game_length=8,
               noise_eps=0.25, tau_decay_rate=0.98)
    cfg.model.cnn_filter_num, cfg.model.res_layer_num = 32, 2
    cfg.engine.games_per_gpu = 16
    cfg.engine.report_every_rounds = 16
    cfg.play_data.max_file_num = 1000
    model = CChessModel(cfg)
    model.build(seed=0)
    w = SelfPlayWorker(cfg, model=model)
    totals = w.run(max_rounds=4000, max_games=24)
    w.close()
    assert totals["games"] >= 24 and totals["expansions"] > 0
    assert totals["red_wins"] + totals["black_wins"] + totals["draws"] == totals["games"]
    files = get_game_data_filenames(cfg.resource)
    assert len(files) == w.stored_games and len(files) > 0
    for path in files:
        data = read_game_data_from_file(path)
        assert data[0] == xo.INIT_STATE
        state, vals = data[0], []
        for mv, val in data[1:]:
            assert mv in xo.get_legal_moves(state)            # every recorded move is playable
            state = xo.step(state, mv)
            vals.append(val)
        assert all(v in (-1, 0, 1) for v in vals)
        assert all(vals[i] == -vals[i - 1] for i in range(1, len(vals)))   # alternating sign (self_play.py:202-208)
        assert len(vals) <= 2 * cfg.play.max_game_length + 1


def test_record_decoder_matches_oracle_replay(tmp_path, monkeypatch):
    """GPU form of the trainer's expanding_data (optimize.py:234-281) vs an oracle replay."""
    import torch
    from cchess_alphazero.lib.record_decoder import expand_records, split_games
    rng = np.random.default_rng(3)
    games = []
    for g in range(7):
        state, data = xo.INIT_STATE, [xo.INIT_STATE]
        for ply in range(int(rng.integers(1, 40))):
            if xo.done(state)[0]:
                break
            mv = xo.get_legal_moves(state)
            m = mv[int(rng.integers(len(mv)))]
            data.append([m, 1 if ply % 2 == 0 else -1])
            state = xo.step(state, m)
        games.append(data)
    flat = [x for g in games for x in g]
    assert split_games(flat) == games
    planes, pol, vals, off = expand_records(games)
    planes, pol, vals = planes.cpu().numpy(), pol.cpu().numpy(), vals.cpu().numpy()
    k = 0
    for g in games:
        state = g[0]
        for mv, v in g[1:]:
            assert (planes[k] == xo.state_to_planes(state)).all()
            assert pol[k] == xo.label_of_str(mv) and vals[k] == v
            state = xo.step(state, mv)
            k += 1
    assert k == len(planes) == off[-1]
    with pytest.raises(ValueError):
        expand_records([[xo.INIT_STATE, ['4445', 1]]])


def _arena_cfg(tmp_path, monkeypatch, gm, K=1):
    cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=gm["sims"], search_threads=K, noise_eps=0.0,
               tau_decay_rate=0.0, c_puct=gm.get("c_puct", 1.0), max_game_length=gm["max_game_length"])
    cfg.opts.evaluate = bool(gm.get("evaluate", False))
    return cfg


def test_evaluator_arena_reproduces_reference_games(tmp_path, monkeypatch):
    """EvaluateWorker.play_games against games recorded from the REFERENCE's own
    EvaluateWorker.start_game (tests/golden/arena_k1.json; generator make_golden_mcts.py arena): result, length and,
    ply by ply, the position searched, the move, the root visit counts, the banned moves and the temperature flag."""
    from arena_oracle import visit_crc
    from cchess_alphazero.worker.evaluator import EvaluateWorker
    with open(os.path.join(GOLDEN, "arena_k1.json")) as f:
        games = json.load(f)["games"]
    for gm in games:
        cfg = _arena_cfg(tmp_path, monkeypatch, gm)
        evs = tuple((lambda planes, s=x: stub_net.hash_stub_torch(planes, s)) for x in gm["salts"])
        w = EvaluateWorker(cfg, evaluators=evs, seed=5)
        trace = {}
        got = w.play_games(1, u_fn=lambda g, turns, _s=gm["seed"]: stub_net.philox_uniform(_s, g, 1, turns),
                           indices=[gm["idx"]], init_state=gm.get("init_state"), trace=trace)
        assert got == [(gm["value"], gm["turns"])], gm["name"]
        tr = trace[gm["idx"]]
        assert len(tr) == len(gm["plies"]), gm["name"]
        for t, r in zip(tr, gm["plies"]):
            assert t["state"] == r["state"] and t["action"] == r["action"], (gm["name"], t, r)
            assert visit_crc(t["moves"], t["n"]) == r["crc"] and t["sum_n"] == r["sum_n"], gm["name"]
            assert t["no_act"] == r["no_act"] and t["inc"] == r["inc"], gm["name"]


@pytest.mark.parametrize("K,sims,tau", [(4, 24, 0.9), (8, 40, 0.0)])
def test_evaluator_arena_matches_oracle(tmp_path, monkeypatch, K, sims, tau):
    """Two models, two trees per game, colours alternating by game index, all games concurrent, K > 1: against the
    arena loop over two oracle players (tests/arena_oracle.py, itself pinned to the reference's games at K = 1)."""
    from arena_oracle import arena_game
    from cchess_alphazero.worker.evaluator import EvaluateWorker, score_table
    cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=sims, search_threads=K, noise_eps=0.0,
               tau_decay_rate=tau, c_puct=1.0, max_game_length=14)
    cfg.opts.evaluate = False
    specs = (dict(kind="hash", salt=41), dict(kind="hash", salt=42))
    evs = tuple((lambda planes, s=s: stub_net.hash_stub_torch(planes, s["salt"])) for s in specs)

    def u_fn(g, turns):
        return stub_net.philox_uniform(99, g, 1, turns)
    n = 10
    w = EvaluateWorker(cfg, evaluators=evs, seed=5)
    got = w.play_games(n, u_fn=u_fn)
    exp = [arena_game(i, cfg.play, specs, u_fn)[:2] for i in range(n)]
    assert got == exp
    table = score_table(got)
    assert sum(table[1:]) == n and 0 <= table[0] <= n


def test_evaluator_arena_on_the_compact_queue(tmp_path, monkeypatch):
    """The arena with two real inference networks takes the compact-queue path (cz_search_round_q + cz_*_q: leaf rows
    and their count stay on the device, one completion check every other round): it plays the same number of
    simulations per ply as the slot-queue path and every game ends with a legal result."""
    import torch
    from cchess_alphazero import _native
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    from cchess_alphazero.worker.evaluator import EvaluateWorker, score_table
    cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=24, search_threads=8, noise_eps=0.0, tau_decay_rate=0.0,
               c_puct=1.0, max_game_length=6)
    nets = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        nets.append(InferenceNet(CChessNet(cnn_filter_num=128, res_layer_num=1), torch.float32, trunk="mfma").cuda())
    out = {}
    for compact in (True, False):
        w = EvaluateWorker(cfg, evaluators=tuple(nets), dtype=_native.U8, seed=3)
        assert w.compact_capable and not w.compact           # (available, off by default: slower at this size)
        w.compact = compact
        stats = {}
        res = w.play_games(6, u_fn=lambda g, t: 0.5, stats=stats)
        assert len(res) == 6 and all(v in (-1, 0, 1) and 0 < t <= 13 for v, t in res)
        assert stats["overflow_sims"] == 0 and stats["tree_resets"] == 0
        out[compact] = (res, stats["sims"], stats["plies"])
        assert sum(score_table(res)[1:]) == 6
    # same games either way (the networks see the same positions; only the batch composition differs)
    assert out[True][2] == out[False][2] and out[True][1] == out[False][1]
    assert out[True][0] == out[False][0]


def test_run_py_self_cli(tmp_path):
    """`python cchess_alphazero/run.py self --type mini ...` (reference CLI) end to end: play records appear."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "chinesechess-alphazero_amd")
    env = dict(os.environ, DATA_DIR=str(tmp_path / "data"), PROJECT_DIR=str(tmp_path), PYTHONPATH=pkg)
    # a small bounded run: the mini type plays 100-sim searches with a 7x256 net; keep it short
    r = subprocess.run([sys.executable, os.path.join(pkg, "cchess_alphazero", "run.py"), "self", "--type", "mini",
                        "--games-per-gpu", "64", "--max-rounds", "60"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.isdir(tmp_path / "data" / "play_data") and os.path.exists(tmp_path / "logs" / "play.log")


def test_run_py_eval_cli(tmp_path):
    """`python cchess_alphazero/run.py eval --type mini` (reference CLI, manager.py:94-103): BestModel vs
    NextGenerationModel (no files: two random-init networks) through the arena worker; the score table is logged."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "chinesechess-alphazero_amd")
    env = dict(os.environ, DATA_DIR=str(tmp_path / "data"), PROJECT_DIR=str(tmp_path), PYTHONPATH=pkg)
    r = subprocess.run([sys.executable, os.path.join(pkg, "cchess_alphazero", "run.py"), "eval", "--type", "mini"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    log = open(tmp_path / "logs" / "eval.log").read()
    assert "Evaluate over, next generation win" in log and "new\told" in log


def test_inference_net_gpu_matches_fp32_reference():
    """The GPU inference network (BN folded, channels-last, hand-written bias+skip+ReLU epilogue) against the plain
    PyTorch fp32 module on the CPU: policy / value within 1e-4 (north_star tolerance); the fused epilogue against
    PyTorch's own separate passes: equal to rounding."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    torch.manual_seed(3)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    net.eval()
    boards = np.stack([xo.state_to_board(xo.INIT_STATE)] * 3 + [xo.state_to_board('3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4')] * 2)
    x = torch.from_numpy(np.stack([xo.planes_board(b) for b in boards]))
    with torch.no_grad():
        p_ref, v_ref = net(x)
    inf = InferenceNet(net, torch.float32).cuda()
    p1, v1 = inf(x.cuda())
    inf.fused_epilogue = False
    p2, v2 = inf(x.cuda())
    assert (p1 - p2).abs().max() < 1e-6 and (v1 - v2).abs().max() < 1e-5      # same math, different kernels
    assert (p1.cpu() - p_ref).abs().max() < 1e-4 and (v1.cpu() - v_ref).abs().max() < 1e-4
    for dt, tol in ((torch.bfloat16, 3e-2), (torch.float16, 5e-3)):
        lo = InferenceNet(net, dt).cuda()
        p3, v3 = lo(x.cuda())
        lo.fused_epilogue = False
        p4, v4 = lo(x.cuda())
        assert (p3 - p4).abs().max() < tol and (v3 - v4).abs().max() < tol * 10
        assert (v3.cpu() - v_ref).abs().max() < tol * 10


def test_uci_front_end_depth_infinite_stop():
    """The UCI front-end on the engine (reference uci.py): `go depth`, `go infinite` + `stop`,
    `position ... moves`, `bestmove .. ponder ..` in the GUI's frame, legal moves only."""
    import io
    import time
    from cchess_alphazero.config import Config, PlayWithHumanConfig
    from cchess_alphazero.uci import UCI
    cfg = Config(config_type="mini")
    PlayWithHumanConfig().update_play_config(cfg.play)
    cfg.play.search_threads = 8
    cfg.resource.model_best_config_path = "/nonexistent/config.json"      # random-init network
    out = io.StringIO()
    u = UCI(cfg, out=out)
    for line in ("uci", "isready", "ucinewgame", "position startpos moves h2e2 h9g7"):
        u.handle(line)
    assert "uciok" in out.getvalue() and "readyok" in out.getvalue()
    assert u.turns == 2 and u.is_red_turn and len(u.history) == 5

    def wait_bestmove(n, timeout=120):
        t0 = time.time()
        while out.getvalue().count("bestmove") < n:
            assert time.time() - t0 < timeout, out.getvalue()
            time.sleep(0.05)
        return [l for l in out.getvalue().splitlines() if l.startswith("bestmove")][-1]

    u.handle("go depth 2")                                  # 200 simulations
    best = wait_bestmove(1)
    tok = best.split()
    assert tok[0] == "bestmove" and len(tok[1]) == 4
    red_moves = {senv_to_uci(m) for m in _legal(u.state)}
    assert tok[1] in red_moves
    info = [l for l in out.getvalue().splitlines() if l.startswith("info depth")]
    assert len(info) >= 2 and " pv " in info[0] and "nps" in info[-1]      # per-depth lines carry the PV
    if len(tok) == 4:
        assert tok[2] == "ponder" and len(tok[3]) == 4

    # black to move: moves come back flipped into the GUI's frame
    u.handle("position startpos moves h2e2")
    assert not u.is_red_turn and u.turns == 1
    u.handle("go infinite")
    time.sleep(1.0)
    u.handle("stop")
    best2 = wait_bestmove(2)
    from cchess_alphazero.environment.lookup_tables import flip_move
    black_moves = {senv_to_uci(flip_move(m)) for m in _legal(u.state)}
    assert best2.split()[1] in black_moves
    assert out.getvalue().count("bestmove") == 2           # one answer per go, also when stop races the search
    u.handle("go movetime 300")
    wait_bestmove(3)
    assert u.handle("quit") is False


def _legal(state):
    from cchess_alphazero.environment import static_env as senv
    return senv.get_legal_moves(state)


def senv_to_uci(m):
    from cchess_alphazero.environment import static_env as senv
    return senv.to_uci_move(m)


def test_model_api_hot_reload(tmp_path):
    """CChessModelAPI.try_reload_model (reference agent/api.py:76-87): a new best-weight file on disk replaces the
    network that serves the pipes; an unchanged file does not."""
    import torch
    from cchess_alphazero.agent.model import CChessModel
    from cchess_alphazero.config import Config
    from cchess_alphazero.lib import model_helper
    cfg = Config(config_type="mini")
    cfg.resource.model_best_config_path = str(tmp_path / "model_best_config.json")
    cfg.resource.model_best_weight_path = str(tmp_path / "model_best_weight.h5")
    m = CChessModel(cfg)
    m.build(seed=1)
    model_helper.save_as_best_model(m)
    pipe = m.get_pipes(need_reload=True)
    x = torch.from_numpy(np.stack([xo.planes_board(xo.state_to_board(xo.INIT_STATE))])).cuda()
    p0, v0 = pipe.evaluate_device(x)
    assert not m.api.try_reload_model()
    other = CChessModel(cfg)
    other.build(seed=2)
    model_helper.save_as_best_model(other)
    assert m.api.try_reload_model() and m.digest == other.digest
    p1, v1 = pipe.evaluate_device(x)
    assert (p1 - p0).abs().max() > 1e-6                    # the new weights are the ones being served
    with torch.no_grad():
        pr, vr = other.model.eval()(x.cpu())
    assert (p1.cpu() - pr).abs().max() < 1e-4 and (v1.cpu() - vr).abs().max() < 1e-4


def test_selfplay_worker_reloads_a_new_best_model(tmp_path, monkeypatch):
    """The reference's self-play picks up a new best model while it runs (get_pipes(need_reload=True) -> the prediction
    thread re-checks the best-weight digest every 600 s, agent/api.py:37-44).  SelfPlayWorker.run does the same check
    on its report boundary: after the file on disk changes, the engine's games are played with the new weights
    (also when the rounds are replayed from a HIP graph); an unchanged file changes nothing."""
    import torch
    from cchess_alphazero.agent.model import CChessModel
    from cchess_alphazero.lib import model_helper
    from cchess_alphazero.worker.self_play import SelfPlayWorker
    cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=16, search_threads=4, max_game_length=6, noise_eps=0.0)
    cfg.model.cnn_filter_num, cfg.model.res_layer_num = 32, 2
    cfg.engine.games_per_gpu, cfg.engine.report_every_rounds, cfg.engine.reload_seconds = 8, 4, 0
    cfg.resource.create_directories()
    m = CChessModel(cfg)
    m.build(seed=1)
    model_helper.save_as_best_model(m)
    w = SelfPlayWorker(cfg, model=m)
    w.run(max_rounds=8)
    assert not w.reload_best_model()                         # same digest: nothing to do
    x = w.engine.queue_planes(8)
    p0, _ = w.engine.net(x)
    other = CChessModel(cfg)
    other.build(seed=2)
    model_helper.save_as_best_model(other)
    w.run(max_rounds=4)                                      # the report boundary inside run() reloads
    assert m.digest == other.digest
    p1, v1 = w.engine.net(x)
    assert (p1 - p0).abs().max() > 1e-6
    with torch.no_grad():
        pr, vr = other.model.eval()(x.float().cpu())
    assert (p1.cpu() - pr).abs().max() < 1e-4 and (v1.cpu() - vr).abs().max() < 1e-4
    c = w.run(max_rounds=8)
    assert c["expansions"] > 0
    w.close()


def test_engine_keeps_its_network_when_a_reload_fails_and_audits_live_positions(tmp_path, monkeypatch):
    """(a) a hot reload whose guard raises (out of memory in the float64 calibration, a damaged file) must leave the
    games on the OLD network and its arithmetic -- set_network builds and measures the new network before it touches the
    engine; (b) the running arithmetic is re-measured on LIVE queue positions (engine.audit_network; SelfPlayWorker.audit
    runs it early and every audit_every_rounds) and a failing audit steps the request down one arithmetic."""
    import torch
    from cchess_alphazero import engine as engine_mod
    from cchess_alphazero.agent.model import CChessModel
    from cchess_alphazero.worker.self_play import SelfPlayWorker
    cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=16, search_threads=4, max_game_length=12, noise_eps=0.0)
    cfg.model.cnn_filter_num, cfg.model.res_layer_num = 128, 2
    cfg.engine.games_per_gpu, cfg.engine.report_every_rounds, cfg.engine.reload_seconds = 16, 4, None
    cfg.engine.audit_first_round, cfg.engine.audit_every_rounds = 6, 1000
    m = CChessModel(cfg)
    m.build(seed=1)
    w = SelfPlayWorker(cfg, model=m)
    audits = []
    real_audit = SelfPlayWorker.audit
    monkeypatch.setattr(SelfPlayWorker, "audit", lambda self: audits.append(real_audit(self)) or audits[-1])
    w.run(max_rounds=8)
    assert len(audits) == 1 and audits[0]["ok"] and audits[0]["arith"] == w.engine.net_arith_effective == "c6"
    assert audits[0]["logit_max_abs"] < 2e-4 and audits[0]["policy_max_abs"] < 5e-5
    # (a) a guard that raises: nothing changes
    old_net, x = w.engine.net, w.engine.queue_planes(8)
    p0, _ = old_net(x)
    other = CChessModel(cfg)
    other.build(seed=2)

    def boom(*a, **k):
        raise RuntimeError("HIP out of memory (simulated)")
    monkeypatch.setattr(engine_mod, "guarded_inference_net", boom)
    with pytest.raises(RuntimeError):
        w.engine.set_network(other.model)
    assert w.engine.net is old_net and w.engine.net_arith_effective == "c6"
    assert torch.equal(w.engine.net(x)[0], p0)
    c = w.run(max_rounds=4)                                  # the games go on
    assert c["expansions"] > 0
    monkeypatch.undo()
    # (b) a failing audit: the request steps down -- one reduced block fewer -- and the network is rebuilt through the guard
    monkeypatch.setattr(type(w.engine), "audit_network", lambda self, n=64: {"ok": False, "arith": self.net_arith_effective})
    w.audit()
    assert w.engine.arith == "c6>1" and w.engine.net_arith_effective == "c6>1"
    w.audit()
    assert w.engine.arith == "c8" and w.engine.net_arith_effective == "c8"
    with torch.no_grad():
        pr, vr = m.model.eval()(x.float().cpu())
    p1, v1 = w.engine.net(x)
    assert (p1.cpu() - pr).abs().max() < 1e-4 and (v1.cpu() - vr).abs().max() < 1e-4
    w.close()


def test_uci_searches_match_reference_player(tmp_path, monkeypatch):
    """action(depth=...), the principal variation behind `info depth .. pv ..` and the ponder move against the
    REFERENCE's own player run with uci=True (tests/golden/uci_k1.json, make_golden_uci.py)."""
    import io
    from cchess_alphazero.agent.player import CChessPlayer
    from cchess_alphazero.environment import static_env as senv
    from cchess_alphazero.environment.lookup_tables import flip_move
    with open(os.path.join(GOLDEN, "uci_k1.json")) as f:
        cases = json.load(f)["cases"]
    assert sum(c.get("hist") is not None for c in cases) >= 2
    for c in cases:
        cfg = _cfg(tmp_path, monkeypatch, simulation_num_per_move=800, search_threads=1, noise_eps=0,
                   tau_decay_rate=0, c_puct=1.0)
        pipe = stub_net.StubPipe(lambda p, s=c["salt"]: stub_net.hash_stub_numpy(p, s))
        tree = {}
        # (cases with a `hist` are 28-plane history players: the score of the info line is then the value of the END of
        #  the line evaluated with the planes of the position two plies up the line, player.py:326-333,442-445)
        pl = CChessPlayer(cfg, search_tree=tree, pipes=pipe, enable_resign=False, debugging=True, uci=True,
                          use_history=c.get("hist") is not None, side=c["turns"] % 2)
        pl.out = io.StringIO()
        action, _ = pl.action(c["state"], c["turns"], depth=c["depth"], hist=c.get("hist"))
        assert action == c["action"]
        assert pl.done_tasks == c["done_tasks"]
        lines = [l for l in pl.out.getvalue().splitlines() if l.startswith("info depth")]
        assert lines and int(lines[-1].split()[2]) <= c["final_depth"]
        # the score of the last line: the network value of the END of the principal variation, seen from `side`
        # (player.py:442-445) -- the same position and the same stub network as in the reference run
        last = lines[-1].split()
        if int(last[2]) == c["final_depth"]:
            assert int(last[last.index("score") + 1]) == c["score"], (lines[-1], c["score"])
        pv, t = [], c["turns"]
        for mov in pl.principal_variation(c["state"]):
            pv.append(senv.to_uci_move(flip_move(mov) if t % 2 == 1 else mov))
            t += 1
        assert pv == c["pv"], (pv, c["pv"])
        node = tree.get(senv.step(c["state"], action))
        ponder, cnt = None, 0
        for mov, a in (node.a.items() if node else ()):
            if a.n > cnt:
                ponder, cnt = mov, a.n
        assert ponder == c["ponder"]
        pl.close()
