"""ctypes binding of the batched search object of libczero.so (cz_search_* in include/czero.h)."""
import ctypes as C

import numpy as np

from cchess_alphazero import _native

COUNTER_NAMES = ["sims", "expansions", "terminal_sims", "repetition_sims", "parked", "sum_depth", "max_depth",
                 "edges_visited", "leaf_moves", "plies", "games", "red_wins", "black_wins", "draws", "resigns",
                 "tree_resets", "overflow_sims", "depth_overflow", "root_reused_sims", "ring_dropped", "chunks_taken",
                 "stat_blocks", "no_act_truncated",
                 # only in the CZ_SIM_PROFILE tuning build (the library reports how many counters it has)
                 "cyc_select", "cyc_rules", "cyc_hash", "cyc_expand", "cyc_rep", "cyc_attach", "cyc_resume_load",
                 "cyc_kernel_select", "cyc_kernel_backup"]
MAX_NO_ACT = 32          # csrc/xq_search.h: banned root moves per game and ply


class SearchCfg(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("sims_per_round", C.c_int32), ("simulation_num_per_move", C.c_int32),
                ("virtual_loss", C.c_int32), ("max_nodes_per_game", C.c_int32), ("pool_chunks", C.c_int32),
                ("max_depth", C.c_int32), ("max_game_length", C.c_int32), ("planes_dtype", C.c_int32),
                ("min_resign_turn", C.c_int32), ("evaluate", C.c_int32), ("ring_capacity", C.c_int32),
                ("c_puct", C.c_double), ("noise_eps", C.c_double), ("dirichlet_alpha", C.c_double),
                ("tau_decay_rate", C.c_double), ("resign_threshold", C.c_double), ("enable_resign_rate", C.c_double),
                ("seed", C.c_uint64), ("use_history", C.c_int32), ("reserved", C.c_int32)]


def declare(L):
    vp, i32 = C.c_void_p, C.c_int
    L.cz_search_create.argtypes = [C.POINTER(SearchCfg), C.POINTER(vp)]
    L.cz_search_destroy.argtypes = [vp]
    L.cz_search_bytes.argtypes = [vp]
    L.cz_search_bytes.restype = C.c_size_t
    L.cz_search_info.argtypes = [vp, vp]
    L.cz_search_memory_info.argtypes = [vp, vp, vp]
    L.cz_search_memory_info.restype = i32
    L.cz_search_start_selfplay.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    L.cz_search_set_roots.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cz_search_round.argtypes = [vp, vp, vp, vp, vp]
    L.cz_search_round_q.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.cz_search_round_q.restype = i32
    L.cz_search_reset_trees.argtypes = [vp, vp]
    L.cz_search_set_sims.argtypes = [vp, i32]
    L.cz_search_policy_logits.argtypes = [vp, i32]
    L.cz_search_policy_logits.restype = i32
    L.cz_search_pending.argtypes = [vp, C.POINTER(C.c_int), vp]
    L.cz_search_root_stats.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.cz_search_leaf_rows.argtypes = [vp, vp, vp, C.POINTER(C.c_int), vp]
    L.cz_search_leaf_rows.restype = i32
    L.cz_search_node_stats.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.cz_search_node_stats.restype = i32
    L.cz_search_stop.argtypes = [vp, vp]
    L.cz_search_stop.restype = i32
    L.cz_search_choose.argtypes = [vp, vp, vp, vp]
    L.cz_search_pv.argtypes = [vp, i32, vp, vp, vp]
    L.cz_search_pv.restype = i32
    L.cz_search_counters.argtypes = [vp, vp, vp]
    L.cz_search_game_counters.argtypes = [vp, vp, vp]
    L.cz_search_game_counters.restype = i32
    L.cz_search_drain_records.argtypes = [vp, C.POINTER(C.c_uint), vp, i32, C.POINTER(C.c_int), vp]
    L.cz_debug_sqrt.argtypes = [vp, vp, i32, vp]
    L.cz_debug_noise.argtypes = [C.c_uint64, C.c_uint32, C.c_double, i32, vp, i32, vp]
    L.cz_debug_noise.restype = i32
    for n in ("cz_search_create", "cz_search_destroy", "cz_search_info", "cz_search_start_selfplay",
              "cz_search_set_roots", "cz_search_round", "cz_search_reset_trees", "cz_search_pending",
              "cz_search_root_stats", "cz_search_choose", "cz_search_counters", "cz_search_drain_records",
              "cz_debug_sqrt", "cz_search_set_sims", "cz_search_policy_logits"):
        getattr(L, n).restype = i32


def masks_to_planes(masks, in_planes, dtype=None):
    """Occupancy boards (int32 [n, 96]: word = plane position i * 9 + j, bit c = plane c shows a piece there; what
    cz_search_leaf_masks makes the search kernel write) -> the planes tensor [n, in_planes, 10, 9] they stand for
    (state_to_planes / state_history_to_planes, environment/static_env.py:137-194).  Pure torch: runs on any device."""
    import torch
    n = masks.shape[0]
    c = torch.arange(in_planes, device=masks.device, dtype=torch.int32)
    bits = (masks[:, None, :90].to(torch.int32) >> c[None, :, None]) & 1
    return bits.to(dtype or torch.uint8).view(n, in_planes, 10, 9)


def planes_to_masks(planes):
    """The inverse: planes [n, in_planes, 10, 9] (0 / non-zero) -> occupancy boards int32 [n, 96] (words 90 .. 95 zero)."""
    import torch
    n, c = planes.shape[0], planes.shape[1]
    bits = (planes.reshape(n, c, 90) != 0).to(torch.int64)
    w = (1 << torch.arange(c, device=planes.device, dtype=torch.int64)).view(1, c, 1)
    out = torch.zeros((n, 96), dtype=torch.int64, device=planes.device)
    out[:, :90] = (bits * w).sum(1)
    return out.to(torch.int32)


class Search:
    """Owns one cz_search (device memory for G game trees) plus the evaluation-queue tensors.

    play_config: any object with the reference's ``config.play`` fields (simulation_num_per_move,
    search_threads, c_puct, noise_eps, dirichlet_alpha, tau_decay_rate, virtual_loss, resign_threshold,
    min_resign_turn, max_game_length, enable_resign_rate).
    """

    def __init__(self, play_config, n_games, planes_dtype=_native.F32, evaluate=False, seed=0,
                 max_nodes_per_game=0, pool_chunks=0, max_depth=0, ring_capacity=0, sims_per_round=None,
                 device=None, use_history=False, pool_fraction=None):
        """max_nodes_per_game: sizes a game's hash / chunk table (0 = the whole tree of the longest game);
        pool_chunks: tree memory shared by all games in MiB (0 = what the games can use, at most 80 % of the free
        device memory); pool_fraction (or CZ_POOL_FRACTION in the environment): with pool_chunks = 0, that fraction
        of the currently free device memory instead of 80 % -- for processes that share a GPU (several workers, a
        co-resident trainer, a UCI engine beside self-play; INTEGRATION.md "Memory")."""
        import os
        import torch
        _native.require_gpu()
        self.L = _native.lib()
        if not hasattr(self.L, "cz_search_create"):
            raise _native.NativeError("libczero.so was built without the search kernels")
        pc = play_config
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.G = int(n_games)
        self.K = int(sims_per_round if sims_per_round is not None else pc.search_threads)
        self.planes_dtype = planes_dtype
        if not pool_chunks:
            frac = pool_fraction if pool_fraction is not None else os.environ.get("CZ_POOL_FRACTION")
            if frac:
                frac = float(frac)
                if not 0.0 < frac <= 0.95:
                    raise ValueError(f"pool_fraction {frac}: expected 0 < f <= 0.95")
                free_b, _ = torch.cuda.mem_get_info(self.device)
                pool_chunks = max(1, int(free_b * frac) >> 20)      # (the library raises it to the games' floor if needed)
        cfg = SearchCfg(self.G, self.K, int(pc.simulation_num_per_move), int(pc.virtual_loss), int(max_nodes_per_game),
                        int(pool_chunks), int(max_depth), int(pc.max_game_length), int(planes_dtype),
                        int(pc.min_resign_turn), int(bool(evaluate)), int(ring_capacity),
                        float(pc.c_puct), float(pc.noise_eps), float(pc.dirichlet_alpha), float(pc.tau_decay_rate),
                        float(pc.resign_threshold), float(pc.enable_resign_rate), int(seed),
                        int(bool(use_history)), 0)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _native.check(self.L.cz_search_create(C.byref(cfg), C.byref(h)), "cz_search_create")
        self.h = h
        info = (C.c_int32 * 16)()
        _native.check(self.L.cz_search_info(self.h, info), "cz_search_info")
        (self.G, self.K, self.sims, self.pool_chunks, self.max_chunks, self.hash_cap, self.max_depth, self.max_plies,
         self.record_stride, self.ring_cap, self.n_counters, self.in_planes, self.keep_chunks,
         self.max_no_act) = list(info)[:14]
        self.slots = self.G * self.K
        self.planes = torch.zeros((self.slots, self.in_planes, 10, 9), dtype=_native.torch_dtype(planes_dtype), device=self.device)
        self.policy = torch.zeros((self.slots, _native.NLABELS), dtype=torch.float32, device=self.device)
        self.value = torch.zeros((self.slots,), dtype=torch.float32, device=self.device)
        self.masks = None                      # leaf_masks(): [slots, 96] int32 occupancy boards beside the planes
        self.planes_off = False                # leaf_planes(False): new leaves are written as occupancy boards only
        self._cursor = C.c_uint(0)

    # -- lifetime --
    def close(self):
        if getattr(self, "h", None):
            self.L.cz_search_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self):
        return int(self.L.cz_search_bytes(self.h))

    def memory_info(self):
        """Tree memory: the shared chunk pool (1 MiB chunks) and what the games hold / use of it."""
        out = (C.c_int64 * 8)()
        _native.check(self.L.cz_search_memory_info(self.h, out, self._stream()), "cz_search_memory_info")
        keys = ("pool_chunks", "free_chunks", "held_chunks", "held_chunks_max_game", "tree_bytes", "tree_bytes_max_game",
                "nodes", "nodes_max_game")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- modes --
    def start_selfplay(self, seed=0, first_game_id=0, game_id_stride=0):
        _native.check(self.L.cz_search_start_selfplay(self.h, int(seed), int(first_game_id), int(game_id_stride),
                                                      self._stream()), "cz_search_start_selfplay")
        self._cursor = C.c_uint(0)

    def set_roots(self, boards, turns=None, no_act=None, n_no_act=None, increase_temp=None, enable_resign=None,
                  select_mask=None, prev_boards=None, hist_kind=None):
        """boards: int8 [G,90] cuda tensor; the optional arguments are cuda tensors of the documented dtypes."""
        import torch

        def ptr(t, dt):
            if t is None:
                return None
            assert t.is_cuda and t.is_contiguous() and t.dtype == dt, (t.dtype, dt)
            return C.c_void_p(t.data_ptr())
        assert boards.shape == (self.G, 90)
        assert no_act is None or tuple(no_act.shape) == (self.G, MAX_NO_ACT), "no_act: [G, 32] uint16"
        self._keep = (boards, turns, no_act, n_no_act, increase_temp, enable_resign, select_mask, prev_boards,
                      hist_kind)
        _native.check(self.L.cz_search_set_roots(
            self.h, ptr(boards, torch.int8), ptr(turns, torch.int32), ptr(no_act, torch.uint16),
            ptr(n_no_act, torch.uint8), ptr(increase_temp, torch.uint8), ptr(enable_resign, torch.uint8),
            ptr(select_mask, torch.uint8), ptr(prev_boards, torch.int8), ptr(hist_kind, torch.uint8),
            self._stream()), "cz_search_set_roots")

    def set_sims(self, sims):
        _native.check(self.L.cz_search_set_sims(self.h, int(sims)), "cz_search_set_sims")
        self.sims = int(sims)

    def policy_logits(self, on=True):
        """The policy rows given to round() are raw logits (agent/model.py forward(logits=True)): the priors are formed from
        the legal moves' logits alone -- the softmax denominator cancels in the reference's renormalisation."""
        _native.check(self.L.cz_search_policy_logits(self.h, int(bool(on))), "cz_search_policy_logits")

    def leaf_masks(self, on=True):
        """Every new leaf's position is also written as an occupancy board (self.masks [slots, 96] int32: word = plane
        position, bit c = plane c shows a piece there; cz_search_leaf_masks) -- 384 bytes instead of 1260 (2520), which the
        first residual block's fused input layer takes directly (InferenceNet.forward(masks=...), cz_input_resblock_m)."""
        import torch
        if on and self.masks is None:
            self.masks = torch.zeros((self.slots, 96), dtype=torch.int32, device=self.device)
        if not on:
            self.masks = None
            self.planes_off = False            # (the library switches the planes back on with the boards gone)
        ptr = C.c_void_p(self.masks.data_ptr()) if self.masks is not None else None
        self.L.cz_search_leaf_masks.argtypes = [C.c_void_p, C.c_void_p]
        self.L.cz_search_leaf_masks.restype = C.c_int
        _native.check(self.L.cz_search_leaf_masks(self.h, ptr), "cz_search_leaf_masks")

    def leaf_planes(self, on=True):
        """on=False (needs leaf_masks): a new leaf is written as its occupancy board ONLY -- for a network whose input layer
        reads the boards (InferenceNet.takes_masks); self.planes is then not updated, queue_planes() rebuilds rows on request
        (cz_search_leaf_planes)."""
        self.L.cz_search_leaf_planes.argtypes = [C.c_void_p, C.c_int]
        self.L.cz_search_leaf_planes.restype = C.c_int
        _native.check(self.L.cz_search_leaf_planes(self.h, int(bool(on))), "cz_search_leaf_planes")
        self.planes_off = not on

    def queue_planes(self, n=None, rows=None):
        """The first n rows of the evaluation queue -- or the queue slots `rows` (int64 device tensor, e.g. q_rows[:q_count]: the
        slots that hold a leaf of the last round) -- as a planes tensor (a copy), whatever the kernel writes: with the planes
        switched off they are rebuilt from the occupancy boards (plane c at position pos = bit c of word pos; state_to_planes,
        environment/static_env.py:137-156)."""
        if rows is not None:
            if not self.planes_off:
                return self.planes[rows].clone()
            return masks_to_planes(self.masks[rows], self.in_planes, self.planes.dtype)
        n = self.slots if n is None else min(int(n), self.slots)
        if not self.planes_off:
            return self.planes[:n].clone()
        return masks_to_planes(self.masks[:n], self.in_planes, self.planes.dtype)

    def reset_trees(self):
        _native.check(self.L.cz_search_reset_trees(self.h, self._stream()), "cz_search_reset_trees")

    # -- one lock-step round: tree kernel only (the caller runs the network on self.planes) --
    def round(self, compact=False):
        """compact=True (cz_search_round_q): after the round self.q_rows[:self.q_count] lists the queue slots that hold a
        new leaf; the caller writes the network result of planes[q_rows[i]] to policy[i] / value[i].  Use one form
        for the whole life of a search."""
        if not compact:
            _native.check(self.L.cz_search_round(self.h, C.c_void_p(self.policy.data_ptr()),
                                                 C.c_void_p(self.value.data_ptr()), C.c_void_p(self.planes.data_ptr()),
                                                 self._stream()), "cz_search_round")
            return
        if getattr(self, "q_rows", None) is None:
            import torch
            self.q_rows = torch.zeros((self.slots,), dtype=torch.int32, device=self.device)
            self.q_count = torch.zeros((1,), dtype=torch.int32, device=self.device)
        _native.check(self.L.cz_search_round_q(self.h, C.c_void_p(self.policy.data_ptr()),
                                               C.c_void_p(self.value.data_ptr()), C.c_void_p(self.planes.data_ptr()),
                                               C.c_void_p(self.q_rows.data_ptr()), C.c_void_p(self.q_count.data_ptr()),
                                               self._stream()), "cz_search_round_q")

    def pending(self):
        out = C.c_int(0)
        _native.check(self.L.cz_search_pending(self.h, C.byref(out), self._stream()), "cz_search_pending")
        return out.value

    def leaf_rows(self):
        """After round(): (searches still running, int64 cuda tensor of the queue rows that hold a new leaf).
        One stream synchronisation, like pending()."""
        import torch
        if getattr(self, "_rows", None) is None:
            self._rows = torch.empty((self.slots,), dtype=torch.int32, device=self.device)
            self._row_counts = torch.zeros((2,), dtype=torch.int32, device=self.device)
        out = (C.c_int * 2)()
        _native.check(self.L.cz_search_leaf_rows(self.h, C.c_void_p(self._rows.data_ptr()),
                                                 C.c_void_p(self._row_counts.data_ptr()), out, self._stream()),
                      "cz_search_leaf_rows")
        return int(out[0]), self._rows[:int(out[1])].long()

    def run_until_idle(self, evaluate, max_rounds=1000000):
        """external mode: rounds until every search is complete.  evaluate(planes) -> (policy, value)."""
        rounds = 0
        while rounds < max_rounds:
            self.round()
            rounds += 1
            pending, rows = self.leaf_rows()
            if pending == 0:
                return rounds
            if rows.numel():                       # only the queue rows that carry a new leaf
                p, v = evaluate(self.planes.index_select(0, rows))
                self.policy.index_copy_(0, rows, p.float())
                self.value.index_copy_(0, rows, v.float())
        raise RuntimeError("search did not finish")

    def stop(self):
        """No further simulations are started; the next round() backs up the ones in flight."""
        _native.check(self.L.cz_search_stop(self.h, self._stream()), "cz_search_stop")

    def node_stats(self, path=None):
        """Edges of the node reached from each root along path (label indices, [G, L] or [L] for G = 1);
        path None / empty = the root (same as root_stats)."""
        import torch
        if path is None or len(path) == 0:
            return self.root_stats()
        G, M = self.G, _native.MAXMOVES
        pa = np.asarray(path, dtype=np.uint16).reshape(G, -1)
        pt = torch.from_numpy(pa.view(np.int16).copy()).to(self.device)
        moves = torch.empty((G, M), dtype=torch.uint16, device=self.device)
        n = torch.empty((G, M), dtype=torch.int32, device=self.device)
        w = torch.empty((G, M), dtype=torch.float64, device=self.device)
        p = torch.empty((G, M), dtype=torch.float32, device=self.device)
        sum_n = torch.empty((G,), dtype=torch.int32, device=self.device)
        counts = torch.empty((G,), dtype=torch.uint8, device=self.device)
        _native.check(self.L.cz_search_node_stats(self.h, C.c_void_p(pt.data_ptr()), pa.shape[1],
                                                  C.c_void_p(moves.data_ptr()), C.c_void_p(n.data_ptr()),
                                                  C.c_void_p(w.data_ptr()), C.c_void_p(p.data_ptr()),
                                                  C.c_void_p(sum_n.data_ptr()), C.c_void_p(counts.data_ptr()),
                                                  self._stream()), "cz_search_node_stats")
        return dict(moves=moves.cpu().numpy(), n=n.cpu().numpy(), w=w.cpu().numpy(), p=p.cpu().numpy(),
                    sum_n=sum_n.cpu().numpy(), counts=counts.cpu().numpy())

    def root_stats(self):
        import torch
        G, M = self.G, _native.MAXMOVES
        moves = torch.empty((G, M), dtype=torch.uint16, device=self.device)
        n = torch.empty((G, M), dtype=torch.int32, device=self.device)
        w = torch.empty((G, M), dtype=torch.float64, device=self.device)
        p = torch.empty((G, M), dtype=torch.float32, device=self.device)
        sum_n = torch.empty((G,), dtype=torch.int32, device=self.device)
        counts = torch.empty((G,), dtype=torch.uint8, device=self.device)
        _native.check(self.L.cz_search_root_stats(self.h, C.c_void_p(moves.data_ptr()), C.c_void_p(n.data_ptr()),
                                                  C.c_void_p(w.data_ptr()), C.c_void_p(p.data_ptr()),
                                                  C.c_void_p(sum_n.data_ptr()), C.c_void_p(counts.data_ptr()),
                                                  self._stream()), "cz_search_root_stats")
        return dict(moves=moves.cpu().numpy(), n=n.cpu().numpy(), w=w.cpu().numpy(), p=p.cpu().numpy(),
                    sum_n=sum_n.cpu().numpy(), counts=counts.cpu().numpy())

    def pv(self, max_len=20, with_visits=False):
        """Principal variation of every game (one launch, one copy): list of label lists (with_visits: also the visit
        counts of those edges)."""
        import torch
        moves = torch.empty((self.G, max_len), dtype=torch.uint16, device=self.device)
        visits = torch.empty((self.G, max_len), dtype=torch.int32, device=self.device)
        _native.check(self.L.cz_search_pv(self.h, int(max_len), C.c_void_p(moves.data_ptr()),
                                          C.c_void_p(visits.data_ptr()), self._stream()), "cz_search_pv")
        mv = moves.cpu().numpy()
        out = [[int(x) for x in row[row != 0xFFFF]] for row in mv]
        if with_visits:
            vs = visits.cpu().numpy()
            return out, [[int(x) for x in vs[g, :len(out[g])]] for g in range(self.G)]
        return out

    def choose(self, u=None):
        import torch
        action = torch.empty((self.G,), dtype=torch.int32, device=self.device)
        ut = None
        if u is not None:
            ut = torch.as_tensor(np.asarray(u, dtype=np.float64)).to(self.device)
        _native.check(self.L.cz_search_choose(self.h, C.c_void_p(ut.data_ptr()) if ut is not None else None,
                                              C.c_void_p(action.data_ptr()), self._stream()), "cz_search_choose")
        return action.cpu().numpy()

    def counters(self):
        out = (C.c_uint64 * self.n_counters)()
        _native.check(self.L.cz_search_counters(self.h, out, self._stream()), "cz_search_counters")
        return {k: int(out[i]) for i, k in enumerate(COUNTER_NAMES[:self.n_counters])}

    def game_counters(self):
        """The counters per game, before the sum: numpy uint64 [G, n_counters] (columns: COUNTER_NAMES)."""
        import numpy as np
        out = np.zeros((self.G, self.n_counters), dtype=np.uint64)
        _native.check(self.L.cz_search_game_counters(self.h, out.ctypes.data_as(C.c_void_p), self._stream()),
                      "cz_search_game_counters")
        return out

    def drain_records(self, max_records=4096):
        """Finished games since the last call: list of dict(game_id, turns, value, store, resigned, moves[labels])."""
        buf = np.zeros((max_records, self.record_stride), dtype=np.uint8)
        n = C.c_int(0)
        _native.check(self.L.cz_search_drain_records(self.h, C.byref(self._cursor), buf.ctypes.data, max_records,
                                                     C.byref(n), self._stream()), "cz_search_drain_records")
        out = []
        for i in range(n.value):
            hdr = buf[i, :16].view(np.int32)
            turns = int(hdr[1])
            mv = buf[i, 16:16 + 2 * min(turns, self.max_plies + 2)].view(np.uint16)
            out.append(dict(game_id=int(buf[i, :4].view(np.uint32)[0]), turns=turns, value=int(hdr[2]),
                            store=bool(hdr[3] & 1), resigned=bool(hdr[3] & 2), moves=mv.copy()))
        return out


def debug_noise(alpha, n_moves, n, seed=0, game_key=0):
    """n draws of Dirichlet(alpha * 1_{n_moves})[0] from the search kernel's root-noise generator (float64 cuda tensor)."""
    import torch
    _native.require_gpu()
    out = torch.empty((n,), dtype=torch.float64, device="cuda")
    _native.check(_native.lib().cz_debug_noise(int(seed), int(game_key), float(alpha), int(n_moves),
                                               C.c_void_p(out.data_ptr()), int(n),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cz_debug_noise")
    return out


def debug_sqrt(x):
    """x: int32 cuda tensor -> float64 tensor of sqrt(x + 1) as computed by the PUCT kernel."""
    import torch
    y = torch.empty(x.shape, dtype=torch.float64, device=x.device)
    _native.check(_native.lib().cz_debug_sqrt(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cz_debug_sqrt")
    return y
