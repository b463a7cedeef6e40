"""search_threads > 1 in the REFERENCE is a thread race: the same search gives different visit counts from run to
run.  The engine fixes one order ("canonical order").  This records the SPREAD of the reference:
root visit vectors of repeated K = 8 searches (hash-stub network, noise 0) of a few positions, so that a test can check
that the engine's canonical result is no more of an outlier than the reference's own runs are
(tests/test_gpu_search.py::test_canonical_order_lies_inside_the_reference_spread).  Output: kgt1_spread.json.

    python tests/golden/make_golden_kgt1.py [runs]
"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv_saved, sys.argv = sys.argv, [sys.argv[0], "none"]
sys.path.insert(0, HERE)
import make_golden_mcts as m  # noqa: E402
import numpy as np  # noqa: E402

CASES = [
    dict(name="init_k8", state=m.senv.INIT_STATE, sims=200, K=8, salt=81),
    dict(name="mid_k8", state='r1e1s1e1r/4m4/2k1c1k2/p1p1p1p1p/9/2P6/P3P1P1P/1CK1C1K2/9/R1EMSME1R', sims=200, K=8, salt=82),
    dict(name="end_k4", state='3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4', sims=120, K=4, salt=83),
]


def one(job):
    ci, run = job
    c = CASES[ci]
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    cfg = m.make_cfg(c["sims"])
    cfg.play.search_threads = c["K"]
    pipe = m.stub_net.StubPipe(m.stub_fn(dict(kind="hash", salt=c["salt"])))
    pl = m.ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
    pl.action(c["state"], 0)
    node = pl.tree[c["state"]]
    n = [int(node.a[mv].n) if mv in node.a else 0 for mv in node.legal_moves]
    pl.close()
    return ci, run, n, int(node.sum_n), pipe.n_positions


def main():
    runs = int(sys.argv_saved[1]) if len(sys.argv_saved) > 1 else 48
    jobs = [(ci, r) for ci in range(len(CASES)) for r in range(runs)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count())) as pool:
        res = pool.map(one, jobs, chunksize=2)
    out = []
    for ci, c in enumerate(CASES):
        rec = dict(c)
        rec["visits"] = [n for (i, r, n, s, e) in res if i == ci]
        rec["sum_n"] = [s for (i, r, n, s, e) in res if i == ci]
        rec["evals"] = [e for (i, r, n, s, e) in res if i == ci]
        out.append(rec)
        v = np.array(rec["visits"], dtype=np.float64)
        p = v / v.sum(1, keepdims=True)
        tv = 0.5 * np.abs(p - p.mean(0)).sum(1)
        print(c["name"], "runs", len(v), "distinct visit vectors", len({tuple(x) for x in rec["visits"]}),
              "TV to the mean: median %.3f max %.3f" % (np.median(tv), tv.max()), "sum_n", set(rec["sum_n"]))
    with open(os.path.join(HERE, "kgt1_spread.json"), "w") as f:
        json.dump({"meta": m.meta(), "cases": out}, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
