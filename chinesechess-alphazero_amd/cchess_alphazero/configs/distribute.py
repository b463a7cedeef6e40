"""Reference-compatible module path (configs/distribute.py); the values live in configs/_tables.py."""
from cchess_alphazero import config as _c
from cchess_alphazero.configs._tables import TYPES as _T

_t = _T["distribute"]


def _factory(cls, key):
    return lambda: cls(**_t[key])


ModelConfig = _factory(_c.ModelConfig, "model")
PlayConfig = _factory(_c.PlayConfig, "play")
PlayDataConfig = _factory(_c.PlayDataConfig, "play_data")
TrainerConfig = _factory(_c.TrainerConfig, "trainer")
EvaluateConfig = _factory(_c.EvaluateConfig, "eval")
