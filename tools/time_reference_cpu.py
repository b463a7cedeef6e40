#!/usr/bin/env python3
"""Times the UNMODIFIED reference (imported from /root/reference) on this container's CPU, SURVEY 8(d):
CChessPlayer + static_env, `action()` from INIT_STATE at the `normal` search settings (800 simulations, c_puct 1.5,
virtual loss 3), P = os.cpu_count() independent OS processes with their own seeds, >= 10 seeded repeats each,
median / min / max.  Two networks behind the reference's pipe protocol:
  * stub   -- zero-latency hash stub (isolates tree + rules, like BASELINE.md section 2);
  * torch  -- the 7 x 128 ResNet as a plain PyTorch CPU module (1 thread per process) behind a pipe-compatible shim
              (end to end; Keras / TensorFlow are not installable here).
search_threads = 1 (the deterministic mode) and 40 (configs/normal.py:36-37).  time.perf_counter() around
player.action only; model construction and start-up are excluded.  Build container only (the reference does not
exist on the GPU box).  Writes profiles/r02_reference_cpu.json.

    python tools/time_reference_cpu.py [--procs P] [--sims 800] [--repeats 10]
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


def worker(variant, threads, sims, repeats, seed):
    sys.path[:0] = [REF, os.path.join(REF, "cchess_alphazero"), os.path.join(ROOT, "tests"),
                    os.path.join(ROOT, "chinesechess-alphazero_amd")]
    import numpy as np
    import stub_net
    import cchess_alphazero.environment.static_env as senv
    from cchess_alphazero.config import Config
    from cchess_alphazero.agent.player import CChessPlayer
    np.random.seed(seed)
    cfg = Config('mini')
    pc = cfg.play
    pc.simulation_num_per_move, pc.search_threads = sims, threads
    pc.c_puct, pc.virtual_loss, pc.noise_eps, pc.dirichlet_alpha, pc.tau_decay_rate = 1.5, 3, 0.15, 0.2, 0.9
    if variant == "torch":
        import importlib.util
        import torch
        torch.set_num_threads(1)
        # the plain PyTorch module of THIS repo's agent/model.py (pure torch; imported by path: the package name
        # collides with the reference's)
        spec = importlib.util.spec_from_file_location(
            "czero_model", os.path.join(ROOT, "chinesechess-alphazero_amd", "cchess_alphazero", "agent", "model.py"))
        # model.py imports nothing of the engine at module level beyond torch
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        torch.manual_seed(0)
        net = mod.CChessNet(cnn_filter_num=128, res_layer_num=7).eval()

        def fn(planes):
            with torch.no_grad():
                p, v = net(torch.from_numpy(np.asarray(planes, dtype=np.float32)))
            return p.numpy(), v.numpy()
    else:
        def fn(planes, _s=seed):
            return stub_net.hash_stub_numpy(planes, 1 + _s)
    out = []
    for r in range(repeats):
        pipe = stub_net.StubPipe(fn)
        pl = CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
        t0 = time.perf_counter()
        pl.action(senv.INIT_STATE, 0)
        dt = time.perf_counter() - t0
        out.append({"sims_per_s": sims / dt, "expansions_per_s": pipe.n_positions / dt, "seconds": dt,
                    "nn_batches": pipe.n_batches})
        pl.close()
    print(json.dumps(out))


def stats(v):
    return {"median": statistics.median(v), "min": min(v), "max": max(v), "n": len(v)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", nargs=5)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--repeats", type=int, default=10)
    ap.add_argument("--torch-repeats", type=int, default=2,
                    help="repeats per process of the torch-CPU-network variant (P x this >= 10 samples; one 800-sim "
                         "search with a single-threaded 7x128 forward per leaf takes minutes)")
    a = ap.parse_args()
    if a.worker:
        v, t, s, r, seed = a.worker
        return worker(v, int(t), int(s), int(r), int(seed))
    res = {"host": {"cpus": os.cpu_count(),
                    "model": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")},
           "reference": "NeymarL/ChineseChess-AlphaZero @ /root/reference, unmodified, imported in-process",
           "sims_per_move": a.sims, "repeats_per_process": {"stub": a.repeats, "torch": a.torch_repeats}, "runs": []}
    for variant in ("stub", "torch"):
        for threads in (1, 40):
            procs = a.procs
            reps = a.repeats if variant == "stub" else a.torch_repeats
            t0 = time.perf_counter()
            ps = [subprocess.Popen([sys.executable, __file__, "--worker", variant, str(threads), str(a.sims),
                                    str(reps), str(i)], stdout=subprocess.PIPE, text=True) for i in range(procs)]
            outs = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
            wall = time.perf_counter() - t0
            per = [x["sims_per_s"] for o in outs for x in o]
            ex = [x["expansions_per_s"] for o in outs for x in o]
            run = {"network": variant, "search_threads": threads, "processes": procs,
                   "per_process_sims_per_s": stats(per), "per_process_expansions_per_s": stats(ex),
                   "aggregate_sims_per_s_median": statistics.median(per) * procs,
                   "aggregate_expansions_per_s_median": statistics.median(ex) * procs,
                   "games_per_hour_at_150_plies": statistics.median(per) * procs / a.sims / 150.0 * 3600.0,
                   "wall_s": wall}
            res["runs"].append(run)
            print(json.dumps(run), flush=True)
    res["summary"] = {f"{r['network']}_K{r['search_threads']}": {
        "aggregate_expansions_per_s": r["aggregate_expansions_per_s_median"],
        "per_process_sims_per_s": r["per_process_sims_per_s"]} for r in res["runs"]}
    with open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
