"""Stateless Xiangqi rules on the side-to-move-normalised state string -- the reference's
``cchess_alphazero/environment/static_env.py`` API, computed by the HIP engine.

String <-> int8[90] conversion is host bookkeeping; every rule (move generation, terminal and
check detection, plane encoding, perpetual check/chase tests) runs as a gfx950 kernel through the
C-ABI in include/czero.h.  There is no CPU fallback: without libczero.so + a GPU these raise.
The batched forms (``*_batch``) are what the self-play engine uses; the scalar forms exist for
drop-in compatibility and cost one kernel launch per call.
"""
import numpy as np

from cchess_alphazero import _native
from cchess_alphazero.environment.lookup_tables import ActionLabelsRed, Fen_2_Idx, flip_move, label_index

INIT_STATE = 'rkemsmekr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RKEMSMEKR'
BOARD_HEIGHT = 10
BOARD_WIDTH = 9

_TYPE = {'p': 1, 'c': 2, 'r': 3, 'k': 4, 'e': 5, 'm': 6, 's': 7}
_LETTER = '.pcrkems'


# ---- host bookkeeping: state string <-> int8[90] ----------------------------------------------
def state_to_array(state):
    """state string -> np.int8[90] (square y*9+x, y=0 mover's back rank; +t mover, -t opponent)."""
    b = np.zeros(90, dtype=np.int8)
    x, y = 0, 9
    for ch in state:
        if ch == ' ':
            break
        if ch == '/':
            x, y = 0, y - 1
        elif '1' <= ch <= '9':
            x += int(ch)
        else:
            t = _TYPE[ch.lower()]
            b[y * 9 + x] = t if ch.isupper() else -t
            x += 1
    return b


def array_to_state(b):
    rows = []
    for y in range(9, -1, -1):
        row, gap = '', 0
        for x in range(9):
            p = int(b[y * 9 + x])
            if p == 0:
                gap += 1
                continue
            if gap:
                row += str(gap)
                gap = 0
            row += _LETTER[p].upper() if p > 0 else _LETTER[-p]
        if gap:
            row += str(gap)
        rows.append(row)
    return '/'.join(rows)


def _to_device(states):
    import torch
    _native.require_gpu()
    arr = np.stack([state_to_array(s) for s in states])
    return torch.from_numpy(arr).cuda()


def _labels_to_device(actions):
    import torch
    _native.require_gpu()
    return torch.tensor([label_index(a) for a in actions], dtype=torch.int32).to(torch.uint16).cuda()


# ---- batched forms ------------------------------------------------------------------------------
def get_legal_moves_batch(states):
    moves, counts = _native.movegen(_to_device(states))
    moves, counts = moves.cpu().numpy(), counts.cpu().numpy()
    return [[ActionLabelsRed[m] for m in moves[i, :counts[i]]] for i in range(len(states))]


def done_batch(states, need_check=False):
    over, v, fm, ck = (t.cpu().numpy() for t in _native.done(_to_device(states), need_check))
    out = []
    for i, s in enumerate(states):
        f = None if fm[i] == _native.NOMOVE else ActionLabelsRed[fm[i]]
        if need_check and 's' in s and 'S' in s:
            out.append((bool(over[i]), int(v[i]), f, bool(ck[i])))
        else:
            out.append((bool(over[i]), int(v[i]), f))      # the reference's early returns are 3-tuples
    return out


def new_step_batch(states, actions):
    out, ne = _native.step(_to_device(states), _labels_to_device(actions))
    out, ne = out.cpu().numpy(), ne.cpu().numpy()
    res = []
    for i in range(len(states)):
        if ne[i] == 0xFF:
            raise ValueError(f"No chessman in {actions[i]}, state = {states[i]}")
        res.append((array_to_state(out[i]), bool(ne[i])))
    return res


def state_to_planes_batch(states):
    return _native.encode(_to_device(states), _native.F32).cpu().numpy()


# ---- the reference's scalar API ---------------------------------------------------------------------
def done(state, turns=-1, need_check=False):
    return done_batch([state], need_check)[0]


def step(state, action):
    return new_step_batch([state], [action])[0][0]


def new_step(state, action):
    return new_step_batch([state], [action])[0]


def get_legal_moves(state, board=None):
    return get_legal_moves_batch([state])[0]


def state_to_planes(state):
    return state_to_planes_batch([state])[0]


def state_history_to_planes(state, history):
    planes = np.zeros((28, 10, 9), dtype=np.float32)
    states = [state]
    if history and len(history) >= 5:
        states.append(history[-5])
    enc = state_to_planes_batch(states)
    planes[:14] = enc[0]
    if len(states) == 2:
        planes[14:] = enc[1]
    return planes


def will_check_or_catch(ori_state, action):
    r = int(_native.check_or_catch(_to_device([ori_state]), _labels_to_device([action])).cpu()[0])
    if r == 0xFF:
        raise ValueError(f"No chessman in {action}, state = {ori_state}")
    return bool(r)


def be_catched(state, mov):
    return bool(int(_native.be_catched(_to_device([state]), _labels_to_device([mov])).cpu()[0]))


def has_attack_chessman(state):
    return bool(int(_native.has_attack(_to_device([state])).cpu()[0]))


# ---- pure string helpers (no arithmetic worth a kernel) ----------------------------------------------
def fliped_state(state):
    return array_to_state(-state_to_array(state)[::-1])


_S2B = str.maketrans("kKeEmMsS", "nNbBaAkK")
_B2S = str.maketrans("nNbBaAkK", "kKeEmMsS")


def state_to_board(state):
    """10x9 list of board letters, lower-case = side to move (reference :117-135)."""
    arr = state_to_array(state)
    letters = '.pcrnbak'
    return [[(letters[p] if p > 0 else letters[-p].upper()) if p else '.'
             for p in (int(arr[y * 9 + x]) for x in range(9))] for y in range(10)]


def board_to_state(board):
    letters = '.pcrnbak'
    arr = np.zeros(90, dtype=np.int8)
    for y in range(10):
        for x in range(9):
            ch = board[y][x]
            if ch != '.':
                t = letters.index(ch.lower())
                arr[y * 9 + x] = t if ch.islower() else -t
    return array_to_state(arr)


def fen_to_state(fen):
    return fen.split(' ')[0].translate(_B2S)


def flip_fen(fen):
    parts = fen.split(' ')
    rows = parts[0].split('/')
    pos = "/".join(row[::-1].swapcase() for row in reversed(rows))
    return " ".join([pos, 'w' if parts[1] == 'b' else 'b'] + parts[2:6])


def state_to_fen(state, turns):
    fen = state.translate(_S2B) + f' w - - 0 {turns}'
    return fen if turns % 2 == 0 else flip_fen(fen)


def parse_onegreen_move(move):
    return f"{int(move[0])}{9 - int(move[1])}{int(move[2])}{9 - int(move[3])}"


def parse_ucci_move(move):
    return f"{ord(move[0]) - ord('a')}{move[1]}{ord(move[2]) - ord('a')}{move[3]}"


def to_uci_move(action):
    return f"{chr(ord('a') + int(action[0]))}{action[1]}{chr(ord('a') + int(action[2]))}{action[3]}"


def init(pos):
    """onegreen 64-digit position string -> state (reference :359-368)."""
    pieces = 'rnbakabnrccpppppRNBAKABNRCCPPPPP'
    board = [['.'] * 9 for _ in range(10)]
    for k, piece in enumerate(pieces):
        p = pos[2 * k:2 * k + 2]
        if p != '99':
            board[9 - int(p[1])][int(p[0])] = piece
    # `pieces` uses board letters with lower-case = the side at the bottom
    return board_to_state(board)


def render(state):
    from logging import getLogger
    board = state_to_board(state)
    for i in range(9, -1, -1):
        getLogger(__name__).debug(board[i])
