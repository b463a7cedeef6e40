"""oracle/xq_oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper over oracle/libxq_oracle.so (the plain-C restatement of the
reference's rules + MCTS).  Mirrors the string-level API of the reference's
``cchess_alphazero/environment/static_env.py`` so tests can compare it 1:1 with
the golden vectors generated from the reference (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libxq_oracle.so")

NSQ, NLABELS, MAXMOVES, NOMOVE = 90, 2086, 128, 0xFFFF
INIT_STATE = 'rkemsmekr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RKEMSMEKR'


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(('.c', '.h'))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "libxq_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        # XQ_ORACLE_LIB: another build of the same sources (oracle/Makefile `asan`: the sanitizer build the CPU tests run under)
        path = os.environ.get("XQ_ORACLE_LIB")
        if not path:
            build()
            path = _LIB_PATH
        _lib = C.CDLL(path)
        _lib.xqo_init()
        _sig(_lib)
    return _lib


def _sig(L):
    i8p, u16p, u8p, f32p = (C.POINTER(C.c_int8), C.POINTER(C.c_uint16),
                            C.POINTER(C.c_uint8), C.POINTER(C.c_float))
    ip = C.POINTER(C.c_int)
    L.xqo_label_from.argtypes = [C.c_int]; L.xqo_label_from.restype = C.c_int
    L.xqo_label_to.argtypes = [C.c_int]; L.xqo_label_to.restype = C.c_int
    L.xqo_label_of.argtypes = [C.c_int, C.c_int]; L.xqo_label_of.restype = C.c_int
    L.xqo_label_str.argtypes = [C.c_int, C.c_char_p]
    L.xqo_label_parse.argtypes = [C.c_char_p]; L.xqo_label_parse.restype = C.c_int
    L.xqo_flip_label.argtypes = [C.c_int]; L.xqo_flip_label.restype = C.c_int
    L.xqo_state_to_board.argtypes = [C.c_char_p, i8p]; L.xqo_state_to_board.restype = C.c_int
    L.xqo_board_to_state.argtypes = [i8p, C.c_char_p]; L.xqo_board_to_state.restype = C.c_int
    L.xqo_flip_board.argtypes = [i8p, i8p]
    L.xqo_legal_moves.argtypes = [i8p, u16p]; L.xqo_legal_moves.restype = C.c_int
    L.xqo_done.argtypes = [i8p, C.c_int, ip, ip, ip, ip]
    L.xqo_step.argtypes = [i8p, C.c_int, i8p, ip]; L.xqo_step.restype = C.c_int
    L.xqo_planes.argtypes = [i8p, f32p]
    L.xqo_planes_hist.argtypes = [i8p, i8p, f32p]
    L.xqo_will_check_or_catch.argtypes = [i8p, C.c_int]; L.xqo_will_check_or_catch.restype = C.c_int
    L.xqo_be_catched.argtypes = [i8p, C.c_int]; L.xqo_be_catched.restype = C.c_int
    L.xqo_has_attack_chessman.argtypes = [i8p]; L.xqo_has_attack_chessman.restype = C.c_int
    L.xqo_batch_rules.argtypes = [i8p, C.c_int, u16p, u8p, i8p, i8p, u16p, u8p, f32p]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---- tables -----------------------------------------------------------------
def labels():
    L = lib()
    out = []
    buf = C.create_string_buffer(5)
    for i in range(NLABELS):
        L.xqo_label_str(i, buf)
        out.append(buf.value.decode())
    return out


def label_tables():
    """(from[2086], to[2086], label_of[90][90]) as numpy arrays."""
    L = lib()
    fr = np.array([L.xqo_label_from(i) for i in range(NLABELS)], dtype=np.uint8)
    to = np.array([L.xqo_label_to(i) for i in range(NLABELS)], dtype=np.uint8)
    lo = np.full((NSQ, NSQ), NOMOVE, dtype=np.uint16)
    lo[fr, to] = np.arange(NLABELS, dtype=np.uint16)
    return fr, to, lo


def label_of_str(mv):
    r = lib().xqo_label_parse(mv.encode())
    if r < 0:
        raise ValueError(f"not a label: {mv}")
    return r


def label_str(label):
    buf = C.create_string_buffer(5)
    lib().xqo_label_str(int(label), buf)
    return buf.value.decode()


def flip_move(mv):
    return label_str(lib().xqo_flip_label(label_of_str(mv)))


# ---- board-level API ----------------------------------------------------------
def state_to_board(state):
    b = np.zeros(NSQ, dtype=np.int8)
    if lib().xqo_state_to_board(state.encode(), _p(b, C.c_int8)) != 0:
        raise ValueError(f"bad state {state}")
    return b


def board_to_state(board):
    board = np.ascontiguousarray(board, dtype=np.int8)
    buf = C.create_string_buffer(128)
    lib().xqo_board_to_state(_p(board, C.c_int8), buf)
    return buf.value.decode()


def flip_board(board):
    board = np.ascontiguousarray(board, dtype=np.int8)
    out = np.zeros(NSQ, dtype=np.int8)
    lib().xqo_flip_board(_p(board, C.c_int8), _p(out, C.c_int8))
    return out


def legal_moves_board(board):
    board = np.ascontiguousarray(board, dtype=np.int8)
    mv = np.zeros(MAXMOVES, dtype=np.uint16)
    n = lib().xqo_legal_moves(_p(board, C.c_int8), _p(mv, C.c_uint16))
    assert n <= MAXMOVES
    return mv[:n].copy()


def done_board(board, need_check=False):
    board = np.ascontiguousarray(board, dtype=np.int8)
    o, v, f, c = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    lib().xqo_done(_p(board, C.c_int8), int(need_check), C.byref(o), C.byref(v), C.byref(f), C.byref(c))
    return bool(o.value), v.value, f.value, bool(c.value)


def step_board(board, label):
    board = np.ascontiguousarray(board, dtype=np.int8)
    out = np.zeros(NSQ, dtype=np.int8)
    ne = C.c_int()
    if lib().xqo_step(_p(board, C.c_int8), int(label), _p(out, C.c_int8), C.byref(ne)) != 0:
        raise ValueError("No chessman in source square")
    return out, bool(ne.value)


def planes_board(board):
    board = np.ascontiguousarray(board, dtype=np.int8)
    pl = np.zeros((14, 10, 9), dtype=np.float32)
    lib().xqo_planes(_p(board, C.c_int8), _p(pl, C.c_float))
    return pl


def batch_rules(boards, want_planes=True):
    boards = np.ascontiguousarray(boards, dtype=np.int8).reshape(-1, NSQ)
    n = boards.shape[0]
    moves = np.zeros((n, MAXMOVES), dtype=np.uint16)
    counts = np.zeros(n, dtype=np.uint8)
    over = np.zeros(n, dtype=np.int8)
    v = np.zeros(n, dtype=np.int8)
    fm = np.zeros(n, dtype=np.uint16)
    ck = np.zeros(n, dtype=np.uint8)
    planes = np.zeros((n, 14, 10, 9), dtype=np.float32) if want_planes else None
    lib().xqo_batch_rules(_p(boards, C.c_int8), n, _p(moves, C.c_uint16), _p(counts, C.c_uint8),
                          _p(over, C.c_int8), _p(v, C.c_int8), _p(fm, C.c_uint16), _p(ck, C.c_uint8),
                          _p(planes, C.c_float) if want_planes else None)
    return dict(moves=moves, counts=counts, over=over, v=v, final_move=fm, check=ck, planes=planes)


# ---- string-level API (same names/semantics as the reference's static_env) -----
def get_legal_moves(state):
    return [label_str(m) for m in legal_moves_board(state_to_board(state))]


def done(state, need_check=False):
    b = state_to_board(state)
    o, v, f, c = done_board(b, need_check)
    fm = None if f == NOMOVE else label_str(f)
    if not (b == 7).any() or not (b == -7).any():
        return (o, v, fm)          # the reference's early returns are 3-tuples (static_env.py:15-18)
    return (o, v, fm, c) if need_check else (o, v, fm)


def step(state, action):
    out, _ = step_board(state_to_board(state), label_of_str(action))
    return board_to_state(out)


def new_step(state, action):
    out, ne = step_board(state_to_board(state), label_of_str(action))
    return board_to_state(out), ne


def fliped_state(state):
    return board_to_state(flip_board(state_to_board(state)))


def state_to_planes(state):
    return planes_board(state_to_board(state))


def state_history_to_planes(state, history):
    b = state_to_board(state)
    pl = np.zeros((28, 10, 9), dtype=np.float32)
    prev = None
    if history and len(history) >= 5:
        prev = state_to_board(history[-5])
    lib().xqo_planes_hist(_p(b, C.c_int8), _p(prev, C.c_int8) if prev is not None else None,
                          _p(pl, C.c_float))
    return pl


def will_check_or_catch(state, action):
    r = lib().xqo_will_check_or_catch(_p(state_to_board(state), C.c_int8), label_of_str(action))
    if r < 0:
        raise ValueError("No chessman in source square")
    return bool(r)


def be_catched(state, action):
    return bool(lib().xqo_be_catched(_p(state_to_board(state), C.c_int8), label_of_str(action)))


def has_attack_chessman(state):
    return bool(lib().xqo_has_attack_chessman(_p(state_to_board(state), C.c_int8)))


# ---- MCTS / self-play restatement (oracle/xq_mcts.c) ---------------------------------------------
class PlayCfg(C.Structure):
    _fields_ = [("simulation_num_per_move", C.c_int), ("search_threads", C.c_int), ("c_puct", C.c_double),
                ("noise_eps", C.c_double), ("dirichlet_alpha", C.c_double), ("tau_decay_rate", C.c_double),
                ("virtual_loss", C.c_int), ("resign_threshold", C.c_double), ("min_resign_turn", C.c_int),
                ("evaluate", C.c_int), ("max_game_length", C.c_int), ("enable_resign_rate", C.c_double),
                ("use_history", C.c_int)]


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("sims", "expansions", "terminal_sims", "repetition_sims", "parked",
                                          "nn_batches", "nn_positions", "max_depth", "sum_depth",
                                          "sum_edges_visited", "sum_leaf_moves")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float),
                      C.POINTER(C.c_float))
RNG_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int, C.c_uint64)


def play_cfg(simulation_num_per_move=100, search_threads=1, c_puct=1.5, noise_eps=0.0, dirichlet_alpha=0.2,
             tau_decay_rate=0.0, virtual_loss=3, resign_threshold=-0.92, min_resign_turn=20, evaluate=0,
             max_game_length=100, enable_resign_rate=1.0, use_history=0):
    return PlayCfg(simulation_num_per_move, search_threads, c_puct, noise_eps, dirichlet_alpha, tau_decay_rate,
                   virtual_loss, resign_threshold, min_resign_turn, evaluate, max_game_length, enable_resign_rate,
                   use_history)


_mcts_sig_done = False


def _mcts_lib():
    global _mcts_sig_done
    L = lib()
    if not _mcts_sig_done:
        i8p, u16p = C.POINTER(C.c_int8), C.POINTER(C.c_uint16)
        dp = C.POINTER(C.c_double)
        L.xqo_player_create.argtypes = [C.POINTER(PlayCfg), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.xqo_player_create.restype = C.c_void_p
        L.xqo_player_destroy.argtypes = [C.c_void_p]
        L.xqo_player_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.xqo_player_tree_size.argtypes = [C.c_void_p]; L.xqo_player_tree_size.restype = C.c_int
        L.xqo_player_clear_tree.argtypes = [C.c_void_p]; L.xqo_player_clear_tree.restype = None
        L.xqo_player_search.argtypes = [C.c_void_p, i8p, C.c_int, u16p, C.c_int, C.c_int, dp]
        L.xqo_player_search.restype = C.c_int
        L.xqo_player_set_history.argtypes = [C.c_void_p, C.c_int, i8p]
        L.xqo_sample_action.argtypes = [C.POINTER(PlayCfg), dp, C.c_int, C.c_int, C.c_double]
        L.xqo_sample_action.restype = C.c_int
        L.xqo_player_action.argtypes = [C.c_void_p, i8p, C.c_int, u16p, C.c_int, C.c_int, C.c_double, dp]
        L.xqo_player_action.restype = C.c_int
        L.xqo_player_node_stats.argtypes = [C.c_void_p, i8p, u16p, C.POINTER(C.c_int32), dp, C.POINTER(C.c_float),
                                            C.POINTER(C.c_int)]
        L.xqo_player_node_stats.restype = C.c_int
        L.xqo_selfplay_game.argtypes = [C.POINTER(PlayCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u16p,
                                        C.c_int, dp, C.POINTER(C.c_int), C.POINTER(Counters), C.POINTER(C.c_uint32)]
        L.xqo_selfplay_game.restype = C.c_int
        L.xqo_philox_uniform.argtypes = [C.c_void_p, C.c_int, C.c_uint64]; L.xqo_philox_uniform.restype = C.c_double
        _mcts_sig_done = True
    return L


class _Stub:
    """Resolves a stub spec to (function pointer, ctx pointer, keepalive)."""

    def __init__(self, spec):
        L = _mcts_lib()
        if callable(spec):
            def cb(ctx, planes, n, plane_len, policy, value, _f=spec):
                pl = np.ctypeslib.as_array(planes, shape=(n, plane_len // 90, 10, 9))
                p, v = _f(pl)
                np.ctypeslib.as_array(policy, shape=(n, NLABELS))[:] = p
                np.ctypeslib.as_array(value, shape=(n,))[:] = v
            self.keep = EVAL_FN(cb)
            self.fn = C.cast(self.keep, C.c_void_p)
            self.ctx = None
        elif spec["kind"] == "uniform":
            self.keep = C.c_float(spec.get("value", 0.0))
            self.fn = C.cast(L.xqo_stub_uniform, C.c_void_p)
            self.ctx = C.cast(C.pointer(self.keep), C.c_void_p)
        else:
            self.keep = C.c_uint64(spec["salt"])
            self.fn = C.cast(L.xqo_stub_hash, C.c_void_p)
            self.ctx = C.cast(C.pointer(self.keep), C.c_void_p)


class _Rng:
    def __init__(self, seed, game_id):
        L = _mcts_lib()
        self.keep = (C.c_uint64 * 2)(seed, game_id)
        self.fn = C.cast(L.xqo_philox_uniform, C.c_void_p)
        self.ctx = C.cast(self.keep, C.c_void_p)


def philox_uniform(seed, game_id, stream, idx):
    r = _Rng(seed, game_id)
    return _mcts_lib().xqo_philox_uniform(r.ctx, stream, idx)


class Player:
    """C restatement of CChessPlayer (agent/player.py); stub = {'kind': 'uniform'|'hash', ...} or a
    python callable planes[n,14,10,9] -> (policy[n,2086], value[n])."""

    def __init__(self, cfg, stub, enable_resign=False, seed=0, game_id=0):
        self.L = _mcts_lib()
        self.cfg = cfg
        self.stub = _Stub(stub)
        self.rng = _Rng(seed, game_id)
        self.h = self.L.xqo_player_create(C.byref(cfg), int(enable_resign), self.stub.fn, self.stub.ctx,
                                          self.rng.fn, self.rng.ctx)

    def close(self):
        if self.h:
            self.L.xqo_player_destroy(self.h)
            self.h = None

    __del__ = close

    def _na(self, no_act):
        arr = np.array([label_of_str(m) for m in (no_act or [])], dtype=np.uint16)
        return arr, (_p(arr, C.c_uint16) if len(arr) else None)

    def search(self, state, turns=0, no_act=None, increase_temp=False):
        b = state_to_board(state) if isinstance(state, str) else np.ascontiguousarray(state, dtype=np.int8)
        arr, ptr = self._na(no_act)
        pol = np.zeros(NLABELS, dtype=np.float64)
        resign = self.L.xqo_player_search(self.h, _p(b, C.c_int8), turns, ptr, len(arr), int(increase_temp),
                                          _p(pol, C.c_double))
        return bool(resign), pol

    def action(self, state, turns=0, no_act=None, increase_temp=False, u=0.5):
        b = state_to_board(state) if isinstance(state, str) else np.ascontiguousarray(state, dtype=np.int8)
        arr, ptr = self._na(no_act)
        pol = np.zeros(NLABELS, dtype=np.float64)
        a = self.L.xqo_player_action(self.h, _p(b, C.c_int8), turns, ptr, len(arr), int(increase_temp), u,
                                     _p(pol, C.c_double))
        return (None if a < 0 else label_str(a)), pol

    def set_history(self, hist):
        """hist: the `hist` argument of CChessPlayer.action (list alternating state, action, ...), or None."""
        if not hist:
            self.L.xqo_player_set_history(self.h, 0, None)
        elif len(hist) >= 5:
            prev = state_to_board(hist[-5])
            self.L.xqo_player_set_history(self.h, 1, _p(prev, C.c_int8))
        else:
            self.L.xqo_player_set_history(self.h, 2, None)

    def node_stats(self, state):
        b = state_to_board(state) if isinstance(state, str) else np.ascontiguousarray(state, dtype=np.int8)
        mv = np.zeros(MAXMOVES, dtype=np.uint16)
        n = np.zeros(MAXMOVES, dtype=np.int32)
        w = np.zeros(MAXMOVES, dtype=np.float64)
        pr = np.zeros(MAXMOVES, dtype=np.float32)
        sn = C.c_int()
        c = self.L.xqo_player_node_stats(self.h, _p(b, C.c_int8), _p(mv, C.c_uint16), _p(n, C.c_int32),
                                         _p(w, C.c_double), _p(pr, C.c_float), C.byref(sn))
        if c < 0:
            return None
        return dict(moves=mv[:c].copy(), n=n[:c].copy(), w=w[:c].copy(), p=pr[:c].copy(), sum_n=sn.value)

    def counters(self):
        c = Counters()
        self.L.xqo_player_counters(self.h, C.byref(c))
        return c.as_dict()

    def tree_size(self):
        return self.L.xqo_player_tree_size(self.h)

    def clear_tree(self):
        """Forget the tree: replays the engine's pool-exhausted fallback (counter tree_resets)."""
        self.L.xqo_player_clear_tree(self.h)


def sample_action(cfg, policy, turns, increase_temp, u):
    pol = np.ascontiguousarray(policy, dtype=np.float64)
    return _mcts_lib().xqo_sample_action(C.byref(cfg), _p(pol, C.c_double), turns, int(increase_temp), u)


def selfplay_game(cfg, stub, seed, game_id, max_plies=512):
    """SelfPlayWorker.start_game restated; returns dict(moves, value, store, turns, counters, visit_crc)."""
    L = _mcts_lib()
    st, rng = _Stub(stub), _Rng(seed, game_id)
    moves = np.zeros(max_plies + 8, dtype=np.uint16)
    crc = np.zeros(max_plies + 8, dtype=np.uint32)
    value, store, ctr = C.c_double(), C.c_int(), Counters()
    turns = L.xqo_selfplay_game(C.byref(cfg), st.fn, st.ctx, rng.fn, rng.ctx, _p(moves, C.c_uint16), max_plies,
                                C.byref(value), C.byref(store), C.byref(ctr), _p(crc, C.c_uint32))
    return dict(moves=[label_str(m) for m in moves[:turns]], value=value.value, store=bool(store.value),
                turns=turns, counters=ctr.as_dict(), visit_crc=crc.copy())   # one entry per action() call
