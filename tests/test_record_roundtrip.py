"""play records written by the engine (tests/golden/engine_records.json, produced on an MI355X by
tools/make_engine_records.py) are valid input for the reference's trainer.  The oracle replays every record on
any machine; where the reference itself is present (build container) its own ``expanding_data``
(worker/optimize.py:234-281) parses them and must produce the same planes / policies / values."""
import json
import os
import sys

import numpy as np
import pytest

from oracle import xq_oracle as xo

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "engine_records.json")


def _games():
    if not os.path.exists(FIXTURE):
        pytest.skip("engine_records.json not generated yet")
    with open(FIXTURE) as f:
        return json.load(f)["games"]


def test_engine_records_replay_with_oracle():
    games = _games()
    assert len(games) >= 20
    for g in games:
        data = g["data"]
        assert data[0] == xo.INIT_STATE and len(data) - 1 == g["turns"]
        state = data[0]
        for i, (mv, v) in enumerate(data[1:]):
            assert mv in xo.get_legal_moves(state), (g["game_id"], i)
            assert v == (g["value"] if i % 2 == 0 else -g["value"])
            state = xo.step(state, mv)
        over = xo.done(state)[0]
        # a decided game ends in a terminal position: the king capture appended (self_play.py:177-184) or the
        # two kings facing each other (static_env.py:39-49) -- unless it ended by resignation
        if g["value"] != 0 and not g["resigned"]:
            assert over


def test_reference_trainer_parses_engine_records():
    if not os.path.isdir("/root/reference/cchess_alphazero"):
        pytest.skip("reference not present on this machine")
    from unittest.mock import MagicMock
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    try:
        for k in [k for k in sys.modules if k == "cchess_alphazero" or k.startswith("cchess_alphazero.")]:
            del sys.modules[k]
        sys.path[:0] = ["/root/reference", "/root/reference/cchess_alphazero"]
        for name in ("tensorflow", "keras", "keras.engine", "keras.engine.topology", "keras.engine.training",
                     "keras.layers", "keras.layers.convolutional", "keras.layers.core", "keras.layers.merge",
                     "keras.layers.normalization", "keras.regularizers", "keras.backend", "keras.models",
                     "keras.optimizers", "keras.callbacks", "keras.utils", "keras.utils.training_utils"):
            sys.modules.setdefault(name, MagicMock())
        import cchess_alphazero.worker.optimize as opt
        for g in _games():
            out = opt.expanding_data(g["data"])
            assert out is not None
            planes, policy, value = out
            n = g["turns"]
            assert planes.shape == (n, 14, 10, 9) and policy.shape == (n, 2086) and value.shape == (n,)
            state = g["data"][0]
            for i, (mv, v) in enumerate(g["data"][1:]):
                assert (planes[i] == xo.state_to_planes(state)).all()
                assert policy[i].argmax() == xo.label_of_str(mv) and policy[i].sum() == 1
                assert value[i] == v
                state = xo.step(state, mv)
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k not in saved_mods]:
            del sys.modules[k]
        sys.modules.update(saved_mods)
