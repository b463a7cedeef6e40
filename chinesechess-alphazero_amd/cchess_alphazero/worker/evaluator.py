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
from cchess_alphazero._native_search import Search
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
            evaluators = (pipes1.evaluate_device, pipes2.evaluate_device)
        self.evaluators = evaluators
        self.dtype = dtype
        self.seed = seed

    def start(self):
        n = self.config.eval.game_num * max(1, self.config.play.max_processes)
        results = self.play_games(n)
        return score_table(results)

    # ------------------------------------------------------------------------------------------------
    def play_games(self, n_games, u_fn=None):
        """Plays games idx = 0..n_games-1 concurrently; returns [(value from red's view, turns)]."""
        import torch
        _native.require_gpu()
        pc = self.config.play
        G = n_games
        searches = [Search(pc, G, planes_dtype=self.dtype, evaluate=getattr(self.config.opts, "evaluate", True),
                           seed=self.seed + k) for k in range(2)]
        dev = searches[0].device
        boards = torch.from_numpy(np.tile(state_to_array(INIT_STATE), (G, 1))).to(dev)
        hist = [[boards[g].cpu().numpy().copy()] for g in range(G)]       # per game: boards (host)
        acts = [[] for _ in range(G)]                                      # per game: labels
        turns = 0
        live = np.ones(G, dtype=bool)
        value = np.zeros(G, dtype=np.int64)
        game_turns = np.zeros(G, dtype=np.int64)
        final_move = np.full(G, _native.NOMOVE, dtype=np.int64)
        no_eat_count = np.zeros(G, dtype=np.int64)
        check = np.zeros(G, dtype=bool)
        idx = np.arange(G)
        while live.any():
            # -- repetition handling before the move (reference :172-189) --
            no_act = np.full((G, 16), _native.NOMOVE, dtype=np.uint16)
            n_no_act = np.zeros(G, dtype=np.uint8)
            inc = np.zeros(G, dtype=np.uint8)
            q_game, q_move = [], []
            for g in np.nonzero(live & ~check)[0]:
                cur = hist[g][-1]
                for i in range(len(hist[g]) - 1):
                    if (hist[g][i] == cur).all():
                        q_game.append(g)
                        q_move.append(acts[g][i])
            if q_game:
                qb = boards[torch.as_tensor(q_game, device=dev)].contiguous()
                qm = torch.tensor(q_move, dtype=torch.int32, device=dev).to(torch.uint16)
                wcc = _native.check_or_catch(qb, qm).cpu().numpy()
                free = {}
                for g, mv, r in zip(q_game, q_move, wcc):
                    if not live[g]:
                        continue
                    inc[g] = 1
                    if r == 1:
                        if n_no_act[g] < 16:
                            no_act[g, n_no_act[g]] = mv
                            n_no_act[g] += 1
                    else:
                        free[g] = free.get(g, 0) + 1
                        if free[g] >= 3:                                   # idle loop three times: draw
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
            busy = [bool(m.any()) for m in masks]
            while any(busy):
                for k in range(2):
                    if not busy[k]:
                        continue
                    s = searches[k]
                    s.round()
                    if s.pending() == 0:
                        busy[k] = False
                        continue
                    p, v = self.evaluators[k](s.planes)
                    s.policy.copy_(p)
                    s.value.copy_(v)
            u = np.array([u_fn(g, turns) if u_fn else np.random.random_sample() for g in range(G)])
            action = np.full(G, -1, dtype=np.int64)
            for k in range(2):
                if masks[k].any():
                    a = searches[k].choose(u)
                    action[masks[k]] = a[masks[k]]
            # -- apply the moves and the game rules to every live game (reference :196-226) --
            resigned = live & (action < 0)
            value[resigned] = -1
            game_turns[resigned] = turns
            live &= ~resigned
            if not live.any():
                break
            mv = torch.from_numpy(np.where(live, action, 0).astype(np.int32)).to(dev).to(torch.uint16)
            nxt, ne = _native.step(boards, mv)
            ne = ne.cpu().numpy()
            lmask = torch.from_numpy(live).to(dev)
            boards = torch.where(lmask[:, None], nxt, boards).contiguous()
            turns += 1
            no_eat_count = np.where(live, np.where(ne == 1, no_eat_count + 1, 0), no_eat_count)
            over, v, fm, ck = (t.cpu().numpy() for t in _native.done(boards, need_check=True))
            attack = _native.has_attack(boards).cpu().numpy()
            host_boards = boards.cpu().numpy()
            for g in np.nonzero(live)[0]:
                acts[g].append(int(action[g]))
                hist[g].append(host_boards[g].copy())
                game_turns[g] = turns
                if no_eat_count[g] >= 120 or turns >= 2 * pc.max_game_length:
                    live[g] = False
                    value[g] = 0
                    continue
                check[g] = bool(ck[g])
                if over[g]:
                    live[g] = False
                    value[g] = int(v[g])
                    final_move[g] = int(fm[g])
                elif not attack[g]:
                    live[g] = False
                    value[g] = 0
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
