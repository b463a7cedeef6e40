"""Batched self-play engine: thousands of concurrent games on one MI355X.

One engine = one GPU = one process.  Each ROUND is
    cz_search_round (hand-written HIP, one wavefront per game: k_sim(BACKUP) -> k_advance -> k_sim(SELECT);
             backup / select / expand / game rules, leaf planes written straight into the evaluation queue)
 -> one ResNet forward over the whole queue (agent/model.py InferenceNet: hand-written MFMA kernels for the input
    layer fused into the first residual block, the residual tower, the head convolutions, the dense heads)
with no host decision and no device->host copy in between.  Finished games are appended to a device
ring and drained by the host only when it wants to write play-record files.

This replaces the reference's process/thread topology (worker/self_play.py:48-60 ProcessPoolExecutor,
agent/player.py ThreadPoolExecutor + sender/receiver threads, agent/api.py prediction thread + pipes).
"""
import os
import time
from logging import getLogger

import torch

from cchess_alphazero import _native
from cchess_alphazero._native_search import Search
from cchess_alphazero.agent.model import CChessNet, guarded_inference_net
from cchess_alphazero.environment.lookup_tables import ActionLabelsRed
from cchess_alphazero.environment.static_env import INIT_STATE

logger = getLogger(__name__)

_PLANES_CODE = {torch.float32: _native.F32, torch.float16: _native.F16, torch.bfloat16: _native.BF16}


def bytes_per_expansion(mean_depth, mean_edges, mean_leaf_moves):
    """Algorithmic HBM bytes of one node expansion (canonical fp32 accounting):
    13 478 + 14 L + sum_i (16 + 14 C_i) + 28 d."""
    return 13478.0 + 14.0 * mean_leaf_moves + mean_depth * (16.0 + 14.0 * mean_edges) + 28.0 * mean_depth


class SelfPlayEngine:
    def __init__(self, config, n_games, net=None, dtype=torch.float32, device=None, seed=0,
                 max_nodes_per_game=0, pool_chunks=0, max_depth=0, sims_per_round=None, evaluator=None,
                 use_history=False, trunk=None):
        """config: the reference's Config object (config.play.* / config.model.* are read).
        net: a CChessNet (random-init if None).  evaluator: optional callable planes -> (policy, value)
        replacing the network (tests).  trunk: "mfma" (hand-written convolution kernel, the default where the
        filter count allows) or "library" (MIOpen), see agent/model.py InferenceNet."""
        _native.require_gpu()
        self.config = config
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        torch.cuda.set_device(self.device)
        self.dtype = dtype
        self.evaluator = evaluator
        self.net = None
        self.trunk = None
        planes_code = _PLANES_CODE[dtype]
        if evaluator is None:
            if net is None:
                torch.manual_seed(0)
                net = CChessNet.from_model_config(config.model)
            self.model_cfg = net.cfg
            if trunk is None:
                trunk = getattr(getattr(config, "engine", None), "net_trunk", "mfma")
            if net.cfg["cnn_filter_num"] not in (32, 128, 192, 256):
                trunk = "library"
            self.trunk = trunk
            self.arith = os.environ.get("CZ_TOWER_ARITH") or getattr(getattr(config, "engine", None), "net_arith", "bf16x3")
            if trunk == "mfma":
                planes_code = _native.U8      # the hand-written input convolution reads the 0/1 planes as bytes
        self.search = Search(config.play, n_games, planes_dtype=planes_code,
                             evaluate=getattr(config.opts, "evaluate", False), seed=seed,
                             max_nodes_per_game=max_nodes_per_game, pool_chunks=pool_chunks, max_depth=max_depth,
                             sims_per_round=sims_per_round, device=self.device, use_history=use_history,
                             pool_fraction=getattr(getattr(config, "engine", None), "pool_fraction", None))
        self.policy_logits = False             # (an evaluator callable hands over probabilities, like the reference's pipe)
        self._ref_net = net if evaluator is None else None     # (CPU copy of the weights: audit_network's float64 reference)
        if evaluator is None:
            self._install(net)
        # compact evaluation queue: the network runs only on the slots that hold a new leaf (2-7 % of the slots of a
        # sustained self-play round carry none, more with large K); needs the kernels that read the count on the device
        want = getattr(getattr(config, "engine", None), "compact_queue", True)
        self._want_compact = want
        self.compact = bool(want and self.net is not None and self.net.supports_compact_queue())
        self.rounds = 0
        self.seed = seed
        self._graph = None

    def _build_net(self, net):
        """The inference network for these weights with the tower arithmetic checked against float64 (agent/model.py
        guarded_inference_net): engine.net_arith is the REQUEST, .arith_effective what the weights allow.  Nothing of the
        engine changes here: a guard that raises (out of memory, a damaged weight file) leaves the games on the old network."""
        guard = getattr(getattr(self.config, "engine", None), "arith_guard", True)
        return guarded_inference_net(net, self.dtype, trunk=self.trunk, arith=self.arith, device=self.device,
                                     guard=None if guard else False)

    def _install(self, net, built=None):
        self.net = built if built is not None else self._build_net(net)
        self.net_arith_effective = self.net.arith_effective
        self.net_calibration = self.net.calibration
        # the engine's own queue: raw logits instead of a softmax over all 2086 columns (the search spreads the priors over
        # the legal moves anyway, reference player.py:272-283: the denominator cancels) -- engine.policy_logits = False opts out
        want = getattr(getattr(self.config, "engine", None), "policy_logits", True)
        self.policy_logits = bool(want and self.net.supports_logits())
        self.search.policy_logits(self.policy_logits)
        # the leaves' occupancy boards beside their planes: the fused input layer takes them directly (engine.leaf_masks = False
        # opts out; only the hand-written trunk on byte planes reads them)
        want_m = getattr(getattr(self.config, "engine", None), "leaf_masks", True) and os.environ.get("CZ_LEAF_MASKS", "1") != "0"
        use_m = bool(want_m and self.trunk == "mfma" and self.search.planes.dtype == torch.uint8)
        self.search.leaf_masks(use_m)
        # ... and where the network reads NOTHING but the boards (input layer fused into the first block) the kernel stops
        # writing the planes: a leaf costs a code row + two word stores instead of the 1260-element encoder pass
        # (engine.leaf_planes = True or CZ_LEAF_PLANES=1 keeps them; queue_planes() rebuilds rows for the audits either way)
        keep_p = getattr(getattr(self.config, "engine", None), "leaf_planes", False) or os.environ.get("CZ_LEAF_PLANES", "0") == "1"
        if use_m:
            self.search.leaf_planes(bool(keep_p or not self.net.takes_masks(self.search.planes.dtype)))

    # ---- control ----
    def set_network(self, net):
        """Swap the weights the games are played with (hot reload of the best model, reference agent/api.py:76-87):
        rebuilds the inference network; a captured HIP graph holds the old weights' buffers and is captured again.
        The new network is built and measured FIRST (the float64 calibration allocates a few hundred MB beside a
        search pool that owns most of HBM): if that raises, the engine keeps playing on the old weights -- and its graph --
        and the exception goes to the caller."""
        if self.evaluator is not None:
            raise RuntimeError("the engine was built on an evaluator callable, not on a network")
        if net.cfg != self.model_cfg:
            raise ValueError(f"hot reload: topology changed ({self.model_cfg} -> {net.cfg})")
        torch.cuda.synchronize(self.device)
        built = self._build_net(net)                           # may raise: nothing has been touched yet
        had_graph = self._graph is not None
        self._graph = None
        torch.cuda.synchronize(self.device)
        self._install(net, built)
        self._ref_net = net
        self.compact = bool(self._want_compact and self.net.supports_compact_queue())
        if had_graph:
            self.capture_graph()

    def queue_planes(self, n=64):
        """The first n positions of the evaluation queue as planes (a copy) -- rebuilt from the occupancy boards when the
        search kernel writes only those (Search.queue_planes)."""
        return self.search.queue_planes(n)

    def audit_network(self, n=64):
        """Re-measure the running arithmetic on LIVE queue positions (the load-time calibration set is random
        playouts; the positions a search actually asks about are more tactical).  Takes up to n of the positions the last round
        asked about -- the compact queue's rows q_rows[:q_count]; slots that were never written (empty boards) are dropped --,
        evaluates the float64 reference and the running network on them and returns measure_against_reference's figures + `ok`
        (within_guard: policy / value within GUARD_TOL, logits within LOGIT_TOL) + `positions`.  A failing audit is logged; the
        caller decides (worker/self_play.py asks for the next more exact arithmetic and rebuilds the network through the
        load-time guard on its standard calibration set)."""
        from cchess_alphazero.agent.model import measure_against_reference, reference_forward_f64, within_guard
        if self.net is None or getattr(self, "_ref_net", None) is None:
            return None
        rows = None
        if getattr(self.search, "q_rows", None) is not None:
            cnt = min(int(self.search.q_count.item()), int(n))
            if cnt > 0:
                rows = self.search.q_rows[:cnt].long()
        planes = self.search.queue_planes(rows=rows) if rows is not None else self.queue_planes(n)
        if planes.dtype != torch.uint8:
            planes = (planes != 0).to(torch.uint8)
        planes = planes[planes.flatten(1).ne(0).any(1)].contiguous()          # (a slot no leaf was ever written to is an empty board)
        if planes.shape[0] == 0:
            return None
        m = measure_against_reference(self.net, reference_forward_f64(self._ref_net, planes), planes)
        m["ok"] = within_guard(m)
        m["arith"] = self.net_arith_effective
        m["positions"] = int(planes.shape[0])
        if not m["ok"]:
            logger.warning("live-queue audit: arithmetic %s is outside the guard on %d live positions: %s",
                           self.net_arith_effective, planes.shape[0], m)
        return m

    def demote_arith(self):
        """Lower the REQUESTED tower arithmetic one step towards exactness below what is running now (one reduced block
        fewer -- c6 -> c6>N-1 -> ... -> c6>1 -> c8 -> c8>N-1 -> ... -> f16x3 -> bf16x3; agent/model.py next_more_exact); the
        next _build_net / set_network starts its guard there.  False when there is nothing below."""
        from cchess_alphazero.agent.model import next_more_exact
        cur = self.net_arith_effective or "bf16x3"
        nxt = next_more_exact(cur, len(self.net.res)) if cur.split(">")[0] in ("c6", "c8", "f16x3") else None
        if nxt is None:
            return False
        self.arith = nxt
        return True

    def start(self, first_game_id=0, game_id_stride=0):
        self.search.start_selfplay(self.seed, first_game_id, game_id_stride)

    def prewarm(self, max_iters=8, max_seconds=90.0):
        """Run the network on the (idle) queue until its time settles: MIOpen picks / compiles its solvers on
        the first calls of a new shape (seconds, with naive fallback kernels meanwhile).  Initialisation only.
        Returns the list of forward times (s)."""
        times = []
        if self.net is None:
            return times
        import time as _t
        t_begin = _t.perf_counter()
        for _ in range(max_iters):
            torch.cuda.synchronize(self.device)
            t0 = _t.perf_counter()
            self.net(self.search.planes)
            torch.cuda.synchronize(self.device)
            times.append(_t.perf_counter() - t0)
            if len(times) >= 2 and 0.8 * times[-2] < times[-1] < 1.25 * times[-2]:
                break
            if _t.perf_counter() - t_begin > max_seconds:
                break
        return times

    def _round(self):
        self.search.round(compact=self.compact)

    def _forward(self):
        s = self.search
        if self.evaluator is not None:
            p, v = self.evaluator(s.planes)
        else:
            out = (s.policy, s.value)
            if self.compact:
                p, v = self.net(s.planes, rows=s.q_rows, count=s.q_count, out=out, logits=self.policy_logits, masks=s.masks)
            else:
                p, v = self.net(s.planes, out=out, logits=self.policy_logits, masks=s.masks)
            if p is s.policy:                                  # written in place (the hand-written network path)
                return
        s.policy.copy_(p)
        s.value.copy_(v)

    def step(self):
        """One lock-step round for every game: tree kernel, then the network on the evaluation queue."""
        if self._graph is not None:
            self._graph.replay()
        else:
            self._round()
            self._forward()
        self.rounds += 1

    def capture_graph(self, warmup=2):
        """Capture kernel + network forward of one round into a HIP graph (no host work per round)."""
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._round()
                self._forward()
                self.rounds += 1
        torch.cuda.current_stream(self.device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._round()
            self._forward()
        self.rounds += 1          # capture does not execute; the first replay does
        self._graph = g

    def counters(self):
        return self.search.counters()

    def drain(self, max_records=4096):
        """Finished games since the last call, as the reference's play-record lists
        ([init_state, [move, value], ...], self_play.py:202-208) plus metadata."""
        out = []
        for r in self.search.drain_records(max_records):
            v = r["value"]
            data = [INIT_STATE]
            for i, m in enumerate(r["moves"]):
                data.append([ActionLabelsRed[int(m)], v if i % 2 == 0 else -v])
            out.append(dict(game_id=r["game_id"], turns=r["turns"], value=v, store=r["store"],
                            resigned=r["resigned"], data=data))
        return out

    def close(self):
        self.search.close()
