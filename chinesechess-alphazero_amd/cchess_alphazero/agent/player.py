"""``CChessPlayer`` with the reference's constructor / ``action`` / ``close`` contract
(cchess_alphazero/agent/player.py:36-196), backed by the gfx950 search kernels.

One player owns one device search object with a single game tree (external mode of
``cz_search_*``).  Search, virtual loss, transposition table, prior spreading, PUCT, backup, policy
and temperature sampling all run on the GPU; this class only moves the position in and the result out.
The throughput path is ``cchess_alphazero.engine.SelfPlayEngine`` (thousands of games per launch);
this class is the drop-in for callers that drive one game at a time.
"""
from logging import getLogger

import numpy as np

from cchess_alphazero import _native
from cchess_alphazero._native_search import Search
from cchess_alphazero.environment import static_env as senv
from cchess_alphazero.environment.lookup_tables import ActionLabelsRed, label_index

logger = getLogger(__name__)


class ActionState:
    def __init__(self, n=0, w=0.0, p=0.0):
        self.n, self.w, self.p = n, w, p
        self.q = w / n if n else 0


class VisitState:
    def __init__(self):
        self.a = {}
        self.sum_n = 0
        self.legal_moves = None
        self.p = None
        self.waiting = False
        self.visit = []
        self.w = 0


class CChessPlayer:
    def __init__(self, config, search_tree=None, pipes=None, play_config=None, enable_resign=False,
                 debugging=False, uci=False, use_history=False, side=0):
        import torch
        self.config = config
        self.play_config = play_config or self.config.play
        self.labels_n = len(ActionLabelsRed)
        self.labels = ActionLabelsRed
        self.move_lookup = {m: i for i, m in enumerate(self.labels)}
        self.pipe = pipes
        self.tree = search_tree if search_tree is not None else {}
        self.enable_resign = enable_resign
        self.debugging = debugging
        self.uci = uci
        self.side = side
        self.search_results = {}
        self.debug = {}
        self.done_tasks = 0
        self.root_state = None
        self.no_act = None
        self.increase_temp = False
        self.use_history = use_history
        if pipes is None:
            raise ValueError("CChessPlayer needs a pipe to the network (model.get_pipes())")
        pc = self.play_config
        # simulation count from play_config, lock-step batch from config.play (player.py:155-174)
        merged = type("PC", (), dict(vars(pc)))()
        merged.search_threads = self.config.play.search_threads
        dt = _native.F32
        self._search = Search(merged, 1, planes_dtype=dt, evaluate=getattr(config.opts, "evaluate", False),
                              seed=int(np.random.randint(0, 2 ** 31 - 1)),
                              node_capacity=getattr(getattr(config, "engine", None), "node_capacity", 0) or 0,
                              use_history=use_history)
        self._torch = torch

    # -- network access: device fast path, or the reference's pipe protocol --
    def _evaluate(self, planes):
        if hasattr(self.pipe, "evaluate_device"):
            return self.pipe.evaluate_device(planes)
        self.pipe.send(list(planes.float().cpu().numpy()))
        while not self.pipe.poll(0.001):
            pass
        rets = self.pipe.recv()
        t = self._torch
        p = t.from_numpy(np.stack([np.asarray(r[0], dtype=np.float32) for r in rets])).to(planes.device)
        v = t.tensor([float(r[1]) for r in rets], dtype=t.float32, device=planes.device)
        return p, v

    def close(self, wait=True):
        if getattr(self, "_search", None) is not None:
            self._search.close()
            self._search = None

    def action(self, state, turns, no_act=None, depth=None, infinite=False, hist=None, increase_temp=False):
        if infinite:
            raise NotImplementedError("infinite search (UCI `go infinite` + stop) is not built yet (SURVEY 8 f-3)")
        t = self._torch
        s = self._search
        base_sims = int(self.play_config.simulation_num_per_move)
        s.set_sims(int(depth) if depth else base_sims)     # action(depth=...): that many simulations (player.py:160)
        self.root_state, self.no_act, self.increase_temp = state, no_act, increase_temp
        board = t.from_numpy(senv.state_to_array(state)[None]).cuda()
        na = np.full((1, 16), 0xFFFF, dtype=np.uint16)
        bans = list(no_act or [])[:16]
        for k, m in enumerate(bans):
            na[0, k] = label_index(m)
        prev, kind = None, None
        if self.use_history and hist:                      # action(hist=...): player.py:150-151, :217-218
            if len(hist) >= 5:
                prev = t.from_numpy(senv.state_to_array(hist[-5])[None]).cuda()
                kind = t.tensor([1], dtype=t.uint8, device="cuda")
            else:
                kind = t.tensor([2], dtype=t.uint8, device="cuda")
        s.set_roots(board, prev_boards=prev, hist_kind=kind,
                    turns=t.tensor([turns], dtype=t.int32, device="cuda"),
                    no_act=t.from_numpy(na.view(np.int16)).cuda().view(t.uint16),
                    n_no_act=t.tensor([len(bans)], dtype=t.uint8, device="cuda"),
                    increase_temp=t.tensor([1 if increase_temp else 0], dtype=t.uint8, device="cuda"),
                    enable_resign=t.tensor([1 if self.enable_resign else 0], dtype=t.uint8, device="cuda"))
        before = s.counters()["sims"]
        s.run_until_idle(self._evaluate)
        self.done_tasks = s.counters()["sims"] - before
        st = s.root_stats()
        c = int(st["counts"][0])
        policy = np.zeros(self.labels_n)
        node = VisitState()
        node.sum_n = int(st["sum_n"][0])
        node.legal_moves = [self.labels[int(m)] for m in st["moves"][0, :c]]
        for j, mov in enumerate(node.legal_moves):
            n, w, p = int(st["n"][0, j]), float(st["w"][0, j]), float(st["p"][0, j])
            node.a[mov] = ActionState(n, w, p)
            policy[self.move_lookup[mov]] = 0 if (no_act and mov in no_act) else n
        self.tree[state] = node
        if self.debugging:
            order = sorted(node.a.items(), key=lambda kv: -kv[1].n)[:5]
            self.search_results = {m: (a.n, a.q, a.p) for m, a in order}
        action = int(s.choose([float(np.random.random_sample())])[0])
        if action < 0:
            return None, list(policy)                      # resign: un-normalised counts (player.py:189-190)
        total = policy.sum()
        if total > 0:
            policy /= total
        return self.labels[action], list(policy)
