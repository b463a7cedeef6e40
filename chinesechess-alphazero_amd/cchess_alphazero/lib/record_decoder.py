"""Batched decoder of play records into training tensors -- the GPU form of the reference trainer's
``expanding_data`` / ``convert_to_trainging_data`` (cchess_alphazero/worker/optimize.py:234-281):
replay every game with ``senv.step``, encode each visited position into the 14 input planes, one-hot policy
from the played move, value per ply.

All games advance together: ply t of every game is one ``cz_step`` launch (one wavefront per game), and all
positions are encoded by one ``cz_encode`` launch.
"""
import numpy as np

from cchess_alphazero import _native
from cchess_alphazero.environment.lookup_tables import label_index
from cchess_alphazero.environment.static_env import state_to_array


def split_games(data):
    """A record file may hold several games flat-concatenated (``nb_game_in_file`` > 1, self_play.py:215):
    a game starts at every string item."""
    games, cur = [], None
    for item in data:
        if isinstance(item, str):
            cur = [item]
            games.append(cur)
        else:
            cur.append(item)
    return games


def expand_records(games, dtype=_native.F32):
    """games: list of ``[init_state, [move, value], ...]``.  Returns device tensors
    (planes [N,14,10,9], policy_index [N] int64, value [N] float32) and the per-game offsets, positions ordered
    game by game, ply by ply (the order ``expanding_data`` produces)."""
    import torch
    _native.require_gpu()
    n_games = len(games)
    lens = [len(g) - 1 for g in games]
    T = max(lens) if lens else 0
    boards = torch.from_numpy(np.stack([state_to_array(g[0]) for g in games])).cuda()
    moves = np.zeros((n_games, max(T, 1)), dtype=np.int32)
    for i, g in enumerate(games):
        for t, (mv, _) in enumerate(g[1:]):
            moves[i, t] = label_index(mv)
    moves_d = torch.from_numpy(moves).cuda()
    lens_d = torch.tensor(lens, device="cuda")
    all_boards = torch.empty((T, n_games, 90), dtype=torch.int8, device="cuda")
    for t in range(T):
        all_boards[t] = boards
        nxt, ne = _native.step(boards, moves_d[:, t].to(torch.uint16).contiguous())
        bad = (ne == 0xFF) & (lens_d > t)
        if bool(bad.any()):
            i = int(bad.nonzero()[0])
            raise ValueError(f"No chessman in {games[i][1 + t][0]} (game {i}, ply {t})")
        boards = torch.where((lens_d > t)[:, None], nxt, boards)
    keep = (torch.arange(T, device="cuda")[:, None] < lens_d[None, :])            # [T, G]
    order = keep.t().reshape(-1).nonzero().squeeze(1)                              # game-major
    flat = all_boards.permute(1, 0, 2).reshape(-1, 90)[order].contiguous()
    planes = _native.encode(flat, dtype)
    pol = moves_d.reshape(-1)[order].to(torch.int64)
    vals = torch.tensor([v for g in games for _, v in g[1:]], dtype=torch.float32, device="cuda")
    offsets = np.concatenate([[0], np.cumsum(lens)])
    return planes, pol, vals, offsets
