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
