#!/usr/bin/env python3
"""Times the UNMODIFIED reference (imported from /root/reference) on this container's CPU:
CChessPlayer + static_env with a zero-latency stub network (tree + rules only), as BASELINE.md section 3 plans.
Build container only (the reference is not on the GPU box).  Writes profiles/r01_reference_cpu.json.

    python tools/time_reference_cpu.py [--procs P] [--sims 200] [--repeats 5]
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def worker(threads, sims, repeats, seed):
    sys.path[:0] = [REF, os.path.join(REF, "cchess_alphazero"), os.path.join(ROOT, "tests")]
    import numpy as np
    import stub_net
    import cchess_alphazero.environment.static_env as senv
    from cchess_alphazero.config import Config
    from cchess_alphazero.agent.player import CChessPlayer
    np.random.seed(seed)
    cfg = Config('mini')
    cfg.play.simulation_num_per_move = sims
    cfg.play.search_threads = threads
    cfg.play.noise_eps = 0
    out = []
    for r in range(repeats):
        pipe = stub_net.StubPipe(lambda p: stub_net.hash_stub_numpy(p, 1 + r))
        pl = CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
        t0 = time.perf_counter()
        pl.action(senv.INIT_STATE, 0)
        dt = time.perf_counter() - t0
        out.append({"sims_per_s": sims / dt, "expansions_per_s": pipe.n_positions / dt, "seconds": dt})
        pl.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", nargs=4, type=int)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5)
    a = ap.parse_args()
    if a.worker:
        return worker(*a.worker)
    res = {"host": {"cpus": os.cpu_count(), "model": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")},
           "sims_per_move": a.sims, "repeats": a.repeats, "runs": []}
    for threads in (1, 10):
        for procs in (1, a.procs):
            t0 = time.perf_counter()
            ps = [subprocess.Popen([sys.executable, __file__, "--worker", str(threads), str(a.sims), str(a.repeats), str(i)],
                                   stdout=subprocess.PIPE, text=True) for i in range(procs)]
            outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
            wall = time.perf_counter() - t0
            per = [x["sims_per_s"] for o in outs for x in o]
            ex = [x["expansions_per_s"] for o in outs for x in o]
            res["runs"].append({"search_threads": threads, "processes": procs,
                                "per_process_sims_per_s": {"median": statistics.median(per), "min": min(per), "max": max(per)},
                                "aggregate_sims_per_s_median": statistics.median(per) * procs,
                                "aggregate_expansions_per_s_median": statistics.median(ex) * procs, "wall_s": wall})
            print(json.dumps(res["runs"][-1]), flush=True)
    with open(os.path.join(ROOT, "profiles", "r01_reference_cpu.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
