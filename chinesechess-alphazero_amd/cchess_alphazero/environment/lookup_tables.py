"""Policy label tables (reference: cchess_alphazero/environment/lookup_tables.py).

The 2086-move label set is owned by the native engine (compile-time table in csrc/xq_tables.h);
this module exposes it under the reference's names.
"""
from enum import Enum

import numpy as np

from cchess_alphazero import _native

_label_of, _from, _to = _native.label_tables()

# plane index of each state letter (reference lookup_tables.py:27-42)
Fen_2_Idx = {c: i for i, pair in enumerate(("pP", "cC", "rR", "kK", "eE", "mM", "sS")) for c in pair}

Winner = Enum("Winner", "red black draw")


def _fmt(f, t):
    return f"{f % 9}{f // 9}{t % 9}{t // 9}"


ActionLabelsRed = [_fmt(int(f), int(t)) for f, t in zip(_from, _to)]
_index = {m: i for i, m in enumerate(ActionLabelsRed)}


def flip_move(x):
    """Same move seen from the other side of the board (reference :50-56)."""
    return f"{8 - int(x[0])}{9 - int(x[1])}{8 - int(x[2])}{9 - int(x[3])}"


def flip_action_labels(labels):
    return [flip_move(x) for x in labels]


ActionLabelsBlack = flip_action_labels(ActionLabelsRed)
Unflipped_index = [_index[x] for x in ActionLabelsBlack]


def flip_policy(pol):
    return np.asarray([pol[ind] for ind in Unflipped_index])


def label_index(move):
    """4-digit move string -> label index (KeyError if the move is not in the label set)."""
    return _index[move]
