"""EvaluateWorker.start_game (reference worker/evaluator.py:147-250) restated over two ORACLE players (test
infrastructure).  tests/test_oracle_mcts.py pins it to games recorded from the reference's own EvaluateWorker
(tests/golden/arena_k1.json, K = 1); the GPU arena is then compared with it for K > 1, where the reference is racy and
parity is defined by the canonical order."""
import zlib

import numpy as np

from oracle import xq_oracle as xo


def visit_crc(moves, n):
    return zlib.crc32(np.asarray(n, dtype=np.int32).tobytes(),
                      zlib.crc32(np.asarray(moves, dtype=np.uint16).tobytes())) & 0xFFFFFFFF


def arena_game(idx, pc, specs, u_fn, init_state=None, evaluate=False, trace=None):
    """pc: config.play-like object; specs: (stub of the best model, stub of the next-generation model);
    u_fn(idx, ply) -> the uniform draw of np.random.choice.  Returns (value from red's view, turns); `trace` (a list)
    receives one dict per action() call."""
    def ocfg():
        return xo.play_cfg(simulation_num_per_move=pc.simulation_num_per_move, search_threads=pc.search_threads,
                           c_puct=pc.c_puct, noise_eps=0.0, dirichlet_alpha=pc.dirichlet_alpha,
                           tau_decay_rate=pc.tau_decay_rate, virtual_loss=pc.virtual_loss, evaluate=int(evaluate),
                           max_game_length=pc.max_game_length)
    p1, p2 = xo.Player(ocfg(), specs[0]), xo.Player(ocfg(), specs[1])
    red, black = (p1, p2) if idx % 2 == 0 else (p2, p1)            # :160-168
    state = init_state or xo.INIT_STATE
    history = [state]
    value = turns = no_eat_count = 0
    game_over = check = False
    final_move = None
    while not game_over:
        no_act, increase_temp = None, False
        if not check and state in history[:-1]:                    # :172-189 (before the move, no be_catched branch)
            no_act, increase_temp, free = [], True, 0
            for i in range(len(history) - 1):
                if history[i] == state:
                    if xo.will_check_or_catch(state, history[i + 1]):
                        no_act.append(history[i + 1])
                    else:
                        free += 1
                        if free >= 3:
                            game_over, value = True, 0
                            break
        if game_over:
            break
        pl = red if turns % 2 == 0 else black
        action, _ = pl.action(state, turns, no_act, increase_temp, u_fn(idx, turns))
        if trace is not None:
            st = pl.node_stats(state)
            trace.append(dict(state=state, action=action, crc=visit_crc(st["moves"], st["n"]), sum_n=st["sum_n"],
                              no_act=no_act, inc=increase_temp))
        if action is None:
            value = -1
            break
        history.append(action)
        state, no_eat = xo.new_step(state, action)
        turns += 1
        no_eat_count = no_eat_count + 1 if no_eat else 0
        history.append(state)
        if no_eat_count >= 120 or turns / 2 >= pc.max_game_length:  # :212-214
            game_over, value = True, 0
        else:
            game_over, value, final_move, check = xo.done(state, need_check=True)
            if not game_over and not xo.has_attack_chessman(state):
                game_over, value = True, 0
    if final_move:                                                  # :228-233
        turns += 1
        value = -value
    if turns % 2 == 1:
        value = -value
    evals = [p1.counters()["nn_positions"], p2.counters()["nn_positions"]]
    p1.close()
    p2.close()
    return value, turns, evals
