"""``run.py eval`` arena (reference: cchess_alphazero/worker/evaluator.py): BestModel vs NextGenerationModel,
colours alternating by game index, two search trees per game (one per player), score table.

All games of the arena are played concurrently: two device search objects (one per model, one tree per game
each, external mode of ``cz_search_*``), every ply = set the roots of the games whose mover belongs to that
model, run lock-step rounds (tree kernel + that model's forward) until the searches are complete, pick the
moves on the device, and apply the game rules to all boards with the batched rule kernels
(``cz_step / cz_done / cz_has_attack / cz_check_or_catch``).  The per-game logic follows
``EvaluateWorker.start_game`` (reference :147-250), including its variant of the repetition handling (done
BEFORE the move, no ``be_catched`` branch) -- but ``final_move`` is initialised (SURVEY C-6) and the number of
playouts is ``config.play.simulation_num_per_move`` instead of ``randint(8, 12) * 100`` (BASELINE config 4).
"""
import os
from logging import getLogger

import numpy as np

from cchess_alphazero import _native
from cchess_alphazero._native_search import MAX_NO_ACT, Search
from cchess_alphazero.environment.static_env import INIT_STATE, state_to_array

logger = getLogger(__name__)


def load_model(config, config_path, weight_path, seed=None):
    """The model stored at the given paths, or (no files) a random-init one for synthetic arenas."""
    from cchess_alphazero.agent.model import CChessModel
    model = CChessModel(config)
    if not model.load(config_path, weight_path):
        if seed is None:
            return None
        model.build(seed=seed)
    return model


def score_table(results):
    """results: list of (value from red's view, turns) indexed by game idx.  Returns the reference's tuple
    (total_score, red_new_win, red_new_draw, red_new_fail, black_new_win, black_new_draw, black_new_fail)
    where even idx = best model plays red (reference :105-135)."""
    total = 0.0
    rw = rd = rf = bw = bd = bf = 0
    for idx, (value, _) in enumerate(results):
        even = idx % 2 == 0
        if (value == 1 and even) or (value == -1 and not even):        # best model won
            if even:
                bf += 1
            else:
                rf += 1
        elif (value == 1 and not even) or (value == -1 and even):      # next generation won
            if even:
                bw += 1
            else:
                rw += 1
        else:
            if even:
                bd += 1
            else:
                rd += 1
        score = 0 if value == -1 else (1 if value == 1 else 0.5)
        total += (1 - score) if even else score
    return (total, rw, rd, rf, bw, bd, bf)


class EvaluateWorker:
    def __init__(self, config, pipes1=None, pipes2=None, pid=None, evaluators=None, dtype=_native.F32, seed=0):
        """evaluators: (eval_best, eval_next), callables planes -> (policy, value) on the device.  When omitted,
        pipes1 / pipes2 must offer ``evaluate_device`` (what ``CChessModel.get_pipes`` returns here)."""
        self.config = config
        self.pid = pid
        if evaluators is None:
            # (the inference networks themselves when the pipes are this package's DevicePipe: they take the compact
            #  queue; any other pipe is used through evaluate_device)
            # (... but only when that service never swaps its network: a reload-enabled api replaces api.net, and a
            #  net bound here once would go stale -- those pipes are called through predict_device every time)
            def pick(p):
                api = getattr(p, "api", None)
                if api is not None and getattr(api, "net", None) is not None and not getattr(api, "need_reload", True):
                    return api.net
                return p.evaluate_device
            evaluators = tuple(pick(p) for p in (pipes1, pipes2))
        self.evaluators = evaluators
        self.dtype = dtype
        self.seed = seed
        self.concurrent = True         # the two models' searches of a ply on two streams / host threads
        # compact evaluation queue (cz_search_round_q): both evaluators are inference networks whose kernels read the
        # leaf count on the device
        self.compact_capable = all(callable(getattr(e, "supports_compact_queue", None)) and e.supports_compact_queue()
                                   for e in evaluators)
        # ... measured on the 200-game arena it is the slower of the two (694 k expansions/s replayed from one HIP graph
        # per model-round, 702 k launched eagerly, against 744 k for gathered leaf rows): host work is not what limits
        # an arena round -- 200 wavefronts walking K = 32 simulations one after the other are (~1.3 ms per k_sim
        # launch, as long as the forward it feeds) -- while the compact form runs the dense layers, the softmax and the
        # input convolution's grid over all 6400 queue rows of a model.  Off unless asked for.
        self.compact = False

    def start(self):
        n = self.config.eval.game_num * max(1, self.config.play.max_processes)
        results = self.play_games(n)
        return score_table(results)

    def _capture_round(self, s, k, stream):
        """One model-round on the compact queue as a HIP graph: cz_search_round_q + the forward of model k straight into
        the search object's policy / value tensors.  One eager round first (allocations, library warm-up), then the
        capture (which executes nothing)."""
        import torch

        def one_round():
            s.round(compact=True)
            self.evaluators[k](s.planes, rows=s.q_rows, count=s.q_count, out=(s.policy, s.value))
        one_round()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            one_round()
        return g

    # ------------------------------------------------------------------------------------------------
    def play_games(self, n_games, u_fn=None, indices=None, init_state=None, trace=None, stats=None, on_ply=None,
                   sims_per_round=None, stop_after_plies=None):
        """Plays the games idx = 0..n_games-1 (or the given `indices`) concurrently; returns
        [(value from red's view, turns)] in that order.

        All games advance ply by ply together: at every ply the games whose mover is the best model are searched on
        search object 0 and the others on search object 1 (one tree per game in each), both run lock-step rounds --
        tree kernels, then that model's forward over the queue rows that hold a new leaf -- until every search is complete;
        the moves are picked on the device and the game rules are applied to all boards with the batched rule
        kernels.  Per-game bookkeeping is vectorised; the only per-game Python work is the (rare) repeated position.
        u_fn(idx, ply) -> uniform draw of np.random.choice (default: NumPy's global RNG, like the reference);
        init_state: start position (default INIT_STATE); trace: dict filled with idx -> [one dict per searched ply];
        stats: dict that receives the search counters (rounds, expansions, ...); on_ply(ply, counters_fn): called at
        the start of every ply (bench.py times plies with it); sims_per_round: K (default config.play.search_threads);
        stop_after_plies: stop after that many plies (benchmark legs only: the games still running are reported as they stand)."""
        import torch
        _native.require_gpu()
        pc = self.config.play
        idx = np.arange(n_games) if indices is None else np.asarray(list(indices), dtype=np.int64)
        G = len(idx)
        searches = [Search(pc, G, planes_dtype=self.dtype, evaluate=getattr(self.config.opts, "evaluate", False),
                           seed=self.seed + k, sims_per_round=sims_per_round) for k in range(2)]

        def counters_now():
            c = [s.counters() for s in searches]
            return {k: c[0][k] + c[1][k] for k in ("sims", "expansions", "tree_resets", "overflow_sims", "sum_depth")}
        dev = searches[0].device
        K = searches[0].K
        max_plies = 2 * int(pc.max_game_length) + 2
        boards = torch.from_numpy(np.tile(state_to_array(init_state or INIT_STATE), (G, 1))).to(dev)
        hist = torch.zeros((max_plies + 1, G, 90), dtype=torch.int8, device=dev)      # position searched at each ply
        acts = torch.zeros((max_plies + 1, G), dtype=torch.int32, device=dev)         # move played at each ply
        turns = 0
        live = np.ones(G, dtype=bool)
        value = np.zeros(G, dtype=np.int64)
        game_turns = np.zeros(G, dtype=np.int64)
        final_move = np.full(G, _native.NOMOVE, dtype=np.int64)
        no_eat_count = np.zeros(G, dtype=np.int64)
        check = np.zeros(G, dtype=bool)
        rounds = rows_evaluated = 0
        from concurrent.futures import ThreadPoolExecutor
        streams = [torch.cuda.Stream(dev) for _ in range(2)]
        graphs, warm = [None, None], {}
        pool = ThreadPoolExecutor(max_workers=2)
        if trace is not None:
            from cchess_alphazero.environment.lookup_tables import ActionLabelsRed
            from cchess_alphazero.environment.static_env import array_to_state
            for i in idx:
                trace[int(i)] = []
        while live.any() and (stop_after_plies is None or turns < stop_after_plies):
            if on_ply is not None:
                on_ply(turns, counters_now, rounds)
            hist[turns] = boards
            # -- repetition handling BEFORE the move (reference :172-189; no be_catched branch here) --
            no_act = np.full((G, MAX_NO_ACT), _native.NOMOVE, dtype=np.uint16)
            n_no_act = np.zeros(G, dtype=np.uint8)
            inc = np.zeros(G, dtype=np.uint8)
            bans = {}                                                   # game -> the reference's no_act list
            if turns > 0:
                cand = torch.from_numpy(live & ~check).to(dev)
                eq = (hist[:turns] == boards[None]).all(dim=2) & cand[None]          # [turns, G]
                pairs = eq.t().nonzero().cpu().numpy()                  # (game, earlier ply), ply ascending per game
                if len(pairs):
                    qg = torch.from_numpy(pairs[:, 0]).to(dev)
                    qm = acts[torch.from_numpy(pairs[:, 1]).to(dev), qg]
                    wcc = _native.check_or_catch(boards[qg].contiguous(), qm.to(torch.uint16)).cpu().numpy()
                    qm = qm.cpu().numpy()
                    free = {}
                    for (g, _), mv, r in zip(pairs, qm, wcc):
                        if not live[g]:
                            continue                                    # (the reference's `break` after the draw)
                        inc[g] = 1
                        bans.setdefault(g, [])
                        if r == 1:                                      # the earlier move checks / chases: banned
                            bans[g].append(int(mv))
                            if mv not in no_act[g, :n_no_act[g]]:
                                if n_no_act[g] >= MAX_NO_ACT:
                                    raise RuntimeError(f"game {idx[g]}: more than {MAX_NO_ACT} banned moves")
                                no_act[g, n_no_act[g]] = mv
                                n_no_act[g] += 1
                        else:
                            free[g] = free.get(g, 0) + 1
                            if free[g] >= 3:                            # idle loop three times: draw
                                live[g] = False
                                value[g] = 0
                                game_turns[g] = turns
            if not live.any():
                break
            # -- search: the mover's model; even idx: best = red (reference :160-168) --
            red_is_best = idx % 2 == 0
            mover_is_best = red_is_best if turns % 2 == 0 else ~red_is_best
            masks = [live & mover_is_best, live & ~mover_is_best]
            t_turns = torch.full((G,), turns, dtype=torch.int32, device=dev)
            t_na = torch.from_numpy(no_act.view(np.int16)).to(dev).view(torch.uint16)
            t_nn = torch.from_numpy(n_no_act).to(dev)
            t_inc = torch.from_numpy(inc).to(dev)
            for k in range(2):
                if masks[k].any():
                    searches[k].set_roots(boards, turns=t_turns, no_act=t_na, n_no_act=t_nn, increase_temp=t_inc,
                                          select_mask=torch.from_numpy(masks[k].astype(np.uint8)).to(dev))
            # The two models' searches of this ply are independent: each runs its rounds (tree kernels, leaf-row
            # compaction, forward on the rows that carry a new position) on its own HIP stream from its own host
            # thread, so one model's kernels fill the gaps the other's launch / sync latencies leave.
            def search_ply(k):
                s = searches[k]
                n_rounds = n_rows = 0
                torch.cuda.set_device(dev)                  # (a fresh host thread starts on device 0)
                with torch.cuda.stream(streams[k]):
                    if self.compact:
                        # Compact queue: the network reads the leaf rows and their count on the device, so a whole
                        # model-round (tree kernels, compaction, forward into the queue tensors) has fixed launch
                        # shapes and is replayed from a HIP graph: no host work per round but one graph launch.  A
                        # search of `sims` simulations needs at least sims / K + 1 rounds: the first completion check
                        # (one synchronisation) is made there, then every other round; a round after the searches
                        # are complete finds no leaf and costs almost nothing.
                        first_check = max(1, -(-int(pc.simulation_num_per_move) // K))
                        n_rounds = warm.pop(k, 0)              # (the eager round that preceded this model's capture)
                        while True:
                            graphs[k].replay()
                            n_rounds += 1
                            if n_rounds >= first_check and (n_rounds - first_check) % 2 == 0:
                                n_rows += int(s.q_count.item())          # (sampled: statistics only)
                                if s.pending() == 0:
                                    break
                        return n_rounds, n_rows
                    while True:
                        s.round()
                        n_rounds += 1
                        pending, leaf = s.leaf_rows()      # only the rows that carry a new position are evaluated
                        if pending == 0:
                            break
                        if leaf.numel():
                            p, v = self.evaluators[k](s.planes.index_select(0, leaf))
                            s.policy.index_copy_(0, leaf, p.float())
                            s.value.index_copy_(0, leaf, v.float())
                            n_rows += int(leaf.numel())
                return n_rounds, n_rows
            active = [k for k in range(2) if masks[k].any()]
            main = torch.cuda.current_stream(dev)
            for k in active:
                streams[k].wait_stream(main)
            if self.compact:                                   # captured here, one after the other: the worker threads only replay
                for k in active:
                    if graphs[k] is None:
                        with torch.cuda.stream(streams[k]):
                            graphs[k] = self._capture_round(searches[k], k, streams[k])
                        warm[k] = 1
            if len(active) == 2 and self.concurrent:
                done = list(pool.map(search_ply, active))
            else:
                done = [search_ply(k) for k in active]
            for k in active:
                main.wait_stream(streams[k])
            rounds += sum(d[0] for d in done)
            rows_evaluated += sum(d[1] for d in done)
            u = np.array([u_fn(int(idx[g]), turns) if u_fn else np.random.random_sample() for g in range(G)])
            action = np.full(G, -1, dtype=np.int64)
            for k in range(2):
                if masks[k].any():
                    a = searches[k].choose(u)
                    action[masks[k]] = a[masks[k]]
            if trace is not None:
                st = [searches[k].root_stats() if masks[k].any() else None for k in range(2)]
                cur = boards.cpu().numpy()
                for g in np.nonzero(live)[0]:
                    r = st[0 if masks[0][g] else 1]
                    c = int(r["counts"][g])
                    trace[int(idx[g])].append(dict(
                        state=array_to_state(cur[g]), action=ActionLabelsRed[action[g]] if action[g] >= 0 else None,
                        moves=r["moves"][g, :c].copy(), n=r["n"][g, :c].copy(), sum_n=int(r["sum_n"][g]),
                        no_act=[ActionLabelsRed[m] for m in bans[g]] if g in bans else None, inc=bool(inc[g])))
            # -- apply the moves and the game rules to every live game (reference :196-226) --
            resigned = live & (action < 0)
            value[resigned] = -1
            game_turns[resigned] = turns
            live &= ~resigned
            if not live.any():
                break
            mv32 = torch.from_numpy(np.where(live, action, 0).astype(np.int32)).to(dev)
            acts[turns] = mv32
            nxt, ne = _native.step(boards, mv32.to(torch.uint16))
            lmask = torch.from_numpy(live).to(dev)
            boards = torch.where(lmask[:, None], nxt, boards).contiguous()
            turns += 1
            over, v, fm, ck = _native.done(boards, need_check=True)
            attack = _native.has_attack(boards)
            ne, over, v, fm, ck, attack = (t.cpu().numpy() for t in (ne, over, v, fm, ck, attack))
            no_eat_count = np.where(live, np.where(ne == 1, no_eat_count + 1, 0), no_eat_count)
            game_turns[live] = turns
            capped = live & ((no_eat_count >= 120) | (turns >= 2 * pc.max_game_length))      # :212-214
            value[capped] = 0
            rest = live & ~capped
            check = np.where(rest, ck != 0, check)
            ended = rest & (over != 0)
            value[ended] = v[ended].astype(np.int64)
            final_move[ended] = fm[ended].astype(np.int64)
            bare = rest & ~ended & (attack == 0)                       # neither side can attack: draw (:219-223)
            value[bare] = 0
            live &= ~(capped | ended | bare)
        # -- the king capture is appended, the value turned to red's view (reference :228-241) --
        results = []
        for g in range(G):
            val, t = int(value[g]), int(game_turns[g])
            if final_move[g] != _native.NOMOVE:
                t += 1
                val = -val
            if t % 2 == 1:
                val = -val
            results.append((val, t))
        if stats is not None:
            stats.update(rounds=rounds, plies=turns, games=G, sims_per_round=K, rows_evaluated=rows_evaluated,
                         **counters_now())
            stats["tree_memory"] = [s.memory_info() for s in searches]
        pool.shutdown()
        torch.cuda.synchronize(dev)
        for s in searches:
            s.close()
        return results


def start(config):
    """Entry point of ``run.py eval`` (reference :28-82)."""
    rc = config.resource
    model_bt = load_model(config, rc.model_best_config_path, rc.model_best_weight_path, seed=0)
    model_ng = load_model(config, rc.next_generation_config_path, rc.next_generation_weight_path, seed=1)
    pipes = (model_bt.get_pipes(need_reload=False), model_ng.get_pipes(need_reload=False))
    worker = EvaluateWorker(config, pipes[0], pipes[1], pid=0)
    total, rw, rd, rf, bw, bd, bf = worker.start()
    game_num = config.eval.game_num * max(1, config.play.max_processes)
    logger.info(f"Evaluate over, next generation win {total}/{game_num} = {total * 100 / game_num:.2f}%")
    logger.info("red\tblack\twin\tdraw\tloss")
    logger.info(f"new\told\t{rw}\t{rd}\t{rf}")
    logger.info(f"old\tnew\t{bw}\t{bd}\t{bf}")
    model_bt.close_pipes()
    model_ng.close_pipes()
    return total, (rw, rd, rf, bw, bd, bf)
