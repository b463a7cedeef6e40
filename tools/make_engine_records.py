#!/usr/bin/env python3
"""GPU box: play a few short self-play games with the engine (tiny random net) and dump their play records to
gpurun_out/engine_records.json; committed as tests/golden/engine_records.json they pin the record format
against the reference trainer's parser (tests/test_record_roundtrip.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT]
import torch  # noqa: E402
from cchess_alphazero.config import Config  # noqa: E402
from cchess_alphazero.engine import SelfPlayEngine  # noqa: E402


def main():
    cfg = Config("mini")
    cfg.model.cnn_filter_num, cfg.model.res_layer_num = 32, 2
    cfg.play.simulation_num_per_move, cfg.play.search_threads, cfg.play.max_game_length = 16, 4, 12
    eng = SelfPlayEngine(cfg, 32, seed=11)
    eng.start()
    games = []
    for r in range(20000):
        eng.step()
        if r % 64 == 63:
            games += eng.drain()
            if len(games) >= 40:
                break
    out = [dict(game_id=g["game_id"], turns=g["turns"], value=g["value"], store=g["store"],
                resigned=g["resigned"], data=g["data"]) for g in games[:40]]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "engine_records.json"), "w") as f:
        json.dump({"config": "mini 2x32 net, 16 sims, K=4, max_game_length=12, seed 11", "games": out}, f)
    print("games", len(out), "endings", sorted({(g["value"], g["turns"]) for g in out})[:10], eng.counters())
    eng.close()


if __name__ == "__main__":
    main()
