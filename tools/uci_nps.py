#!/usr/bin/env python3
"""GPU: single-game search latency (the UCI front-end's `nps`, reference agent/player.py:444-448): one CChessPlayer =
one wavefront, `go depth 8` (800 simulations) from the opening position with the 7 x 128 net, for several lock-step
batch sizes K = config.play.search_threads.  Reports simulations per second (the reference prints
int(depth * 100 / duration) * 1000, i.e. 1000 x this) and where a round's time goes (tree kernel launch + sync,
network forward on K rows).  Output: gpurun_out/uci_nps.json.

    python tools/uci_nps.py [--depth 8] [--ks 8,40,64]
"""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--ks", default="8,40,64")
    a = ap.parse_args()
    from cchess_alphazero.agent.model import CChessModel
    from cchess_alphazero.agent.player import CChessPlayer
    from cchess_alphazero.config import Config
    from cchess_alphazero.environment.static_env import INIT_STATE
    os.environ.setdefault("DATA_DIR", "/tmp/uci_nps_data")
    cfg = Config("normal")
    cfg.model.cnn_filter_num, cfg.model.res_layer_num = 128, 7
    cfg.play.noise_eps = 0                      # (the UCI front-end of the reference sets play noise to 0 as well)
    model = CChessModel(cfg)
    model.build(seed=0)
    pipe = model.get_pipes(need_reload=False)
    out = {"net": "7x128 random-init, float32 (split-bf16 trunk)", "position": "INIT_STATE", "runs": []}
    for K in [int(x) for x in a.ks.split(",")]:
        cfg.play.search_threads = K
        best = None
        for rep in range(3):
            pl = CChessPlayer(cfg, search_tree={}, pipes=pipe, enable_resign=False, debugging=True, uci=True, side=0)
            pl.out = io.StringIO()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            action, _ = pl.action(INIT_STATE, 0, depth=a.depth * 100)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c = pl._search.counters()
            lines = [x for x in pl.out.getvalue().splitlines() if x.startswith("info depth")]
            run = {"K": K, "seconds": dt, "sims": c["sims"], "sims_per_s": c["sims"] / dt,
                   "expansions_per_s": c["expansions"] / dt, "uci_nps_field": int(a.depth * 100 / dt) * 1000,
                   "action": action, "last_info": lines[-1] if lines else None, "info_lines": len(lines)}
            pl.close()
            if best is None or run["seconds"] < best["seconds"]:
                best = run
        out["runs"].append(best)
        print(json.dumps(best), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "uci_nps.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
