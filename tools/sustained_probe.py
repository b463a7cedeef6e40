#!/usr/bin/env python3
"""How does a round's time evolve over a long run (clock / thermal behaviour vs game phase)?
Prints, per chunk of rounds: ms per round, mean k_resblock launch (HIP events), mean search-round time (HIP events).

    python tools/sustained_probe.py [--seconds 90] [--chunk 100]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
sys.path.insert(0, ROOT)

import torch

import bench as B
from cchess_alphazero.engine import SelfPlayEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90.0)
    ap.add_argument("--chunk", type=int, default=100)
    ap.add_argument("--node-capacity", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sustained.json"))
    a = ap.parse_args()
    cfg = B.build_config(argparse.Namespace(config="normal", games=None, sims_per_round=None, dtype=None, trunk=None))
    eng = SelfPlayEngine(cfg, cfg.engine.games_per_gpu, seed=20260923, max_nodes_per_game=a.node_capacity)
    eng.start()
    eng.prewarm()
    rows = []
    t_begin = time.perf_counter()
    c_prev = eng.counters()
    while time.perf_counter() - t_begin < a.seconds:
        eng.net.block_events = []
        ev = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.chunk):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng._round()
            e1.record()
            ev.append((e0, e1))
            eng._forward()
            eng.rounds += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = eng.counters()
        blk = eng.net.block_events
        rows.append(dict(t=time.perf_counter() - t_begin, ms_per_round=dt / a.chunk * 1e3,
                         resblock_ms=sum(x.elapsed_time(y) for x, y in blk) / max(1, len(blk)),
                         search_ms=sum(x.elapsed_time(y) for x, y in ev) / len(ev),
                         expansions_per_s=(c["expansions"] - c_prev["expansions"]) / dt,
                         sims_per_s=(c["sims"] - c_prev["sims"]) / dt))
        c_prev = c
        print(json.dumps(rows[-1]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
