"""``run.py self`` worker (reference: cchess_alphazero/worker/self_play.py).

``start(config)`` never returns: it plays self-play games forever and writes play-record JSON files that
the reference's ``opt`` trainer consumes.  Where the reference forks ``max_processes`` Python workers that
each play one game at a time through a pipe to a prediction thread, this worker drives ONE batched engine
per GPU (``config.engine.games_per_gpu`` concurrent games, one wavefront each).  With several GPUs
(``--gpu 0,1,...`` or a torchrun launch) there is one process per GPU with disjoint game ids; the only
collective is an all-reduce of the game counters (RCCL; gloo in the CPU tests).
"""
import os
import time
from logging import getLogger

import numpy as np

from cchess_alphazero.lib.data_helper import PlayDataWriter

logger = getLogger(__name__)

COUNTER_KEYS = ["games", "plies", "sims", "expansions", "red_wins", "black_wins", "draws", "resigns"]


def load_model(config, config_file=None):
    """Best model if its files exist, otherwise a freshly built random-init one (reference :29-46).
    Always 14 input planes (the reference's use_history quirk is not reproduced, SURVEY C-7)."""
    from cchess_alphazero.agent.model import CChessModel
    model = CChessModel(config)
    rc = config.resource
    config_path = rc.model_best_config_path if not config_file else os.path.join(rc.model_dir, config_file)
    if not model.load(config_path, rc.model_best_weight_path):
        model.build(seed=0)
        if os.path.exists(rc.model_best_config_path):
            # a config whose weights are missing (e.g. the reference's Keras JSON without its .h5 blob): leave it alone
            logger.info(f"{rc.model_best_config_path} exists but its weights could not be loaded: playing with a "
                        f"random-init network, nothing is written")
        else:
            try:
                model.save(rc.model_best_config_path, rc.model_best_weight_path)
            except OSError as e:
                logger.info(f"could not save the freshly built model: {e}")
    return model, False


def reduce_counters(counters, group=None):
    """All-reduce (SUM) the game counters over the ranks.  counters: dict name -> int (local).
    Returns the global dict.  One int64[len(COUNTER_KEYS)] message: latency-bound, SURVEY 8(e)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {k: int(counters.get(k, 0)) for k in COUNTER_KEYS}
    # (a single rank under a launcher still all-reduces: the same RCCL path the multi-GPU run takes)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(counters.get(k, 0)) for k in COUNTER_KEYS], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return dict(zip(COUNTER_KEYS, t.tolist()))


def game_id_partition(rank, world, games_per_gpu):
    """Disjoint game ids: slot g of rank r starts at r*G + g and advances by world*G."""
    return rank * games_per_gpu, world * games_per_gpu


class SelfPlayWorker:
    """One GPU's worth of self-play.  ``start()`` loops forever like the reference's worker;
    ``run(max_rounds=..., max_games=...)`` is the bounded form used by tests and benchmarks."""

    def __init__(self, config, pipes=None, pid=None, use_history=False, rank=0, world=1, model=None):
        self.config = config
        self.id = pid
        self.pid = os.getpid()
        self.rank, self.world = rank, world
        self.model = model
        self.engine = None
        self.writer = PlayDataWriter(config, rank, world)
        self.stored_games = 0

    def _make_engine(self):
        import torch
        from cchess_alphazero.engine import SelfPlayEngine
        ec = self.config.engine
        net = self.model.model if self.model is not None else None
        self.engine = SelfPlayEngine(self.config, ec.games_per_gpu, net=net, dtype=getattr(torch, ec.net_dtype),
                                     seed=ec.base_seed, max_nodes_per_game=ec.max_nodes_per_game,
                                     pool_chunks=ec.pool_chunks, max_depth=ec.max_depth,
                                     sims_per_round=ec.sims_per_round)
        first, stride = game_id_partition(self.rank, self.world, ec.games_per_gpu)
        self.engine.start(first, stride)
        if ec.use_hip_graph:
            self.engine.capture_graph()

    def _harvest(self):
        for g in self.engine.drain():
            logger.debug(f"Process {self.pid}-{self.rank} game {g['game_id']} turn={g['turns'] / 2}, "
                         f"winner = {g['value']:.2f} (1 = red, -1 = black, 0 draw)")
            if g["store"]:
                path = self.writer.add_game(g["data"])
                self.stored_games += 1
                if path:
                    logger.info(f"Process {self.pid} save play data to {path}")

    def reload_best_model(self):
        """The reference's self-play picks up a new best model while it runs: its prediction thread re-checks the
        best-weight digest every 600 s (agent/api.py:37-44, get_pipes(need_reload=True) in self_play.py:62-64).
        Same here: when the file on disk differs from the weights in memory, load it and hand it to the engine.
        Games in flight continue with the new weights, as they do in the reference.  Returns True on a reload."""
        from cchess_alphazero.lib.model_helper import load_best_model_weight, need_to_reload_best_model_weight
        if self.model is None or self.engine is None:
            return False
        try:
            if need_to_reload_best_model_weight(self.model) and load_best_model_weight(self.model):
                self.engine.set_network(self.model.model)
                logger.info(f"Process {self.pid}-{self.rank}: best model reloaded, digest {self.model.digest}")
                return True
        except Exception as e:                 # a half-written file: keep playing with the old weights
            logger.error(f"best-model reload failed: {e}")
        return False

    def audit(self):
        """Measure the running arithmetic against float64 on live queue positions; when it is outside the guard there, ask for
        the next more exact arithmetic and rebuild the network through the guard (the games go on with the old one if that
        fails).  Returns the audit's figures (None for an engine built on an evaluator callable)."""
        try:
            m = self.engine.audit_network()
            if m is not None and not m["ok"] and self.engine.demote_arith():
                self.engine.set_network(self.engine._ref_net)
                logger.warning(f"Process {self.pid}-{self.rank}: live audit failed ({m}); tower arithmetic now "
                               f"{self.engine.net_arith_effective}")
            return m
        except Exception as e:                 # an audit must never stop the games
            logger.error(f"live audit failed to run: {e}")
            return None

    def run(self, max_rounds=None, max_games=None):
        if self.engine is None:
            self._make_engine()
        every = max(1, self.config.engine.report_every_rounds)
        reload_s = getattr(self.config.engine, "reload_seconds", 600)
        # live-queue audit of the running tower arithmetic (engine.audit_network): once when the games have left the opening
        # book of identical positions, then every audit_every_rounds; None switches it off
        audit_every = getattr(self.config.engine, "audit_every_rounds", 50000)
        audit_first = getattr(self.config.engine, "audit_first_round", 400)
        t0, r = time.time(), 0
        last_check = t0
        while True:
            self.engine.step()
            r += 1
            if audit_every and (r == audit_first or r % audit_every == 0):
                self.audit()
            if r % every == 0:
                self._harvest()
                if reload_s is not None and time.time() - last_check >= reload_s:
                    last_check = time.time()
                    self.reload_best_model()
                c = reduce_counters(self.engine.counters())
                if self.rank == 0:
                    dt = time.time() - t0
                    lc = self.engine.counters()
                    logger.info(f"rounds={r} games={c['games']} plies={c['plies']} "
                                f"expansions/s={c['expansions'] / dt:.0f} games/h={c['games'] / dt * 3600:.0f} "
                                f"tree_resets={lc['tree_resets']} overflow_sims={lc['overflow_sims']} "
                                f"no_act_truncated={lc.get('no_act_truncated', 0)}")
                if max_games is not None and c["games"] >= max_games:
                    break
            if max_rounds is not None and r >= max_rounds:
                break
        self._harvest()
        return reduce_counters(self.engine.counters())

    def start(self):
        ec = self.config.engine
        return self.run(max_rounds=getattr(ec, "max_rounds", None), max_games=getattr(ec, "max_games", None))

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None


def rendezvous_file():
    """A fresh file for the FileStore rendezvous of the ranks spawned here.  (A TCP port found free and then closed can
    be taken by another process before the ranks bind it -- ADVICE r03; a file in a private temporary directory cannot.)"""
    import atexit
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="czero_rdzv_")
    # the directory lives as long as the process that created it (the parent of the spawned ranks, or the single rank itself):
    # removed when that process exits, so /tmp does not collect one per run
    atexit.register(shutil.rmtree, d, ignore_errors=True)
    return os.path.join(d, "store")


def _rank_main(rank, world, config, store_path):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    devices = [int(x) for x in str(config.opts.device_list).split(",")]
    torch.cuda.set_device(devices[rank] if rank < len(devices) else rank)
    if world > 1 or os.environ.get("CZ_FORCE_DIST") == "1":
        dist.init_process_group("nccl", init_method=f"file://{store_path or rendezvous_file()}", rank=rank,
                                world_size=world)
    model, _ = load_model(config)
    SelfPlayWorker(config, rank=rank, world=world, model=model).start()


def launched_by_torchrun():
    """True when a launcher set up a rendezvous for this process: torch.distributed.run / torchrun (TORCHELASTIC_RUN_ID), or
    RANK + WORLD_SIZE + MASTER_PORT by hand.  A scheduler that merely exports RANK / WORLD_SIZE (no MASTER_PORT) does not
    count: init_process_group could not rendezvous, and the run used to go single-process -- it still does."""
    import torch.distributed as dist
    if not ("WORLD_SIZE" in os.environ and "RANK" in os.environ):
        return False
    return dist.is_torchelastic_launched() or "MASTER_PORT" in os.environ


def start(config):
    """Entry point of ``run.py self`` (reference :48-60)."""
    import torch
    if launched_by_torchrun():                                                # (any world size)
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
        model, _ = load_model(config)
        return SelfPlayWorker(config, rank=rank, world=world, model=model).start()
    devices = str(config.opts.device_list).split(",")
    if len(devices) > 1:
        import torch.multiprocessing as mp
        return mp.spawn(_rank_main, args=(len(devices), config, rendezvous_file()), nprocs=len(devices), join=True)
    return _rank_main(0, 1, config, None)
