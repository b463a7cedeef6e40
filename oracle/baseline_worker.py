#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- CPU baseline of bench.py (cpu_baseline.kind = "port").

Runs the C restatement of the reference's player.py + static_env.py (oracle/xq_mcts.c, xq_rules.c) on P host
cores at once: P independent OS processes (the reference's own topology: one Python worker per process,
worker/self_play.py:55-60), each playing its own self-play games with its own seed, the network replaced by the
hash stub (tree + rules only).  A new player (= a new tree) per game, like SelfPlayWorker.start_game.  Prints one
JSON object: per-process expansions/s and sims/s, to be aggregated by the caller.

    python oracle/baseline_worker.py --procs 256 --seconds 12 --sims 800 --threads 8 --c-puct 1.5 --vl 3
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _one(job):
    seed, a = job
    from oracle import xq_oracle as xo
    ocfg = xo.play_cfg(simulation_num_per_move=a["sims"], search_threads=a["threads"], c_puct=a["c_puct"],
                       noise_eps=0.0, dirichlet_alpha=0.2, tau_decay_rate=0.0, virtual_loss=a["vl"],
                       max_game_length=a["max_game_length"])
    exp = sims = plies = games = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a["seconds"]:
        pl = xo.Player(ocfg, {"kind": "hash", "salt": seed + 1000 * games})
        state, turn = xo.INIT_STATE, 0
        while time.perf_counter() - t0 < a["seconds"]:
            act, _ = pl.action(state, turn, None, False, 0.5)
            if act is None:
                break
            state = xo.step(state, act)
            turn += 1
            plies += 1
            if xo.done(state)[0] or turn >= 2 * a["max_game_length"]:
                break
        c = pl.counters()
        exp += c["expansions"]
        sims += c["sims"]
        games += 1
        pl.close()
    dt = time.perf_counter() - t0
    return {"seed": seed, "expansions_per_s": exp / dt, "sims_per_s": sims / dt, "plies": plies, "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--c-puct", type=float, default=1.5)
    ap.add_argument("--vl", type=int, default=3)
    ap.add_argument("--max-game-length", type=int, default=100)
    a = ap.parse_args()
    procs = a.procs or len(os.sched_getaffinity(0))
    from oracle import xq_oracle as xo
    xo.lib()                                   # build / load once before forking
    args = dict(seconds=a.seconds, sims=a.sims, threads=a.threads, c_puct=a.c_puct, vl=a.vl,
                max_game_length=a.max_game_length)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_one, [(s + 1, args) for s in range(procs)], chunksize=1)
    print(json.dumps({"procs": procs, "wall_seconds": time.perf_counter() - t0, "per_process": res}))


if __name__ == "__main__":
    main()
