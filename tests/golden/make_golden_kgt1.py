"""search_threads > 1 in the REFERENCE is a thread race: the same search gives different visit counts from run to
run.  The engine fixes one order ("canonical order").  This records the SPREAD of the reference
at the production search size: root visit vectors of repeated 800-simulation searches at K = 8 (bench default) and
K = 40 (`configs/normal.py:36-37`) of 12 positions -- the opening, middlegames and endgames of the 1k suite, a mating
position (the proven-win shortcut of `player.py:311-313` fires), a position searched under a ban list (`no_act`,
`player.py:298-300`) -- hash-stub network, noise 0, so that tests can check that the engine's canonical result is no
more of an outlier than the reference's own runs are:
    tests/test_oracle_mcts.py::test_canonical_order_lies_inside_the_reference_spread   (CPU oracle)
    tests/test_gpu_search.py::test_hip_search_lies_inside_the_reference_spread          (HIP engine)
The reference runs with its own thread timing (1 ms sender sleep, 5 ms switch interval: see below).
Output: kgt1_spread.json.

    python tests/golden/make_golden_kgt1.py [runs]        # default 32 runs per (position, K)
"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv_saved, sys.argv = sys.argv, [sys.argv[0], "none"]
sys.path.insert(0, HERE)
import time  # noqa: E402
import make_golden_mcts as m  # noqa: E402
import numpy as np  # noqa: E402

# make_golden_mcts shortens the sender thread's sleep and the interpreter's switch interval (timing only for
# search_threads = 1).  For K > 1 the timing IS the behaviour being recorded -- which thread selects before which
# result arrives -- so both are put back to what an unmodified reference runs with: time.sleep(0.001) in the sender
# (player.py:113-123) and CPython's default 5 ms switch interval.
m.ref_player.sleep = time.sleep
sys.setswitchinterval(0.005)

SIMS = 800
KS = (8, 40)


def _suite_positions():
    """Non-terminal positions of the 1k suite by piece count: (index, state)."""
    with open(os.path.join(HERE, "positions_1k.json")) as f:
        pos = json.load(f)["positions"]
    live = [(i, p["state"], sum(c.isalpha() for c in p["state"])) for i, p in enumerate(pos)
            if not p["done"][0] and len(p["moves"].split()) >= 4]
    full = [x for x in live if x[2] >= 30]
    mid = [x for x in live if 24 <= x[2] <= 29]
    end = [x for x in live if 5 <= x[2] <= 12]
    pick = lambda lst, ks: [lst[(k * len(lst)) // 7 % len(lst)] for k in ks]
    return pick(full, (1, 3, 5)), pick(mid, (2, 4)), pick(end, (1, 3, 5))


def build_cases():
    full, mid, end = _suite_positions()
    base = [dict(name="opening", state=m.senv.INIT_STATE)]
    base += [dict(name="suite%d_full" % i, state=s) for i, s, _ in full]
    base += [dict(name="suite%d_mid" % i, state=s) for i, s, _ in mid]
    base += [dict(name="suite%d_end" % i, state=s) for i, s, _ in end]
    base.append(dict(name="mate_two_rooks", state='4s4/9/9/9/9/9/9/9/3R5/3S1R3'))            # proven wins in the tree
    base.append(dict(name="endgame_rc", state='3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4'))
    base.append(dict(name="mid_banned", state='r1e1s1e1r/4m4/2k1c1k2/p1p1p1p1p/9/2P6/P3P1P1P/1CK1C1K2/9/R1EMSME1R',
                     no_act=['1219', '2214', '4246']))
    cases = []
    for K in KS:
        for j, b in enumerate(base):
            c = dict(b)
            c.update(name="%s_k%d" % (b["name"], K), sims=SIMS, K=K, salt=100 + j)
            cases.append(c)
    return cases


CASES = build_cases()


def one(job):
    ci, run = job
    c = CASES[ci]
    np.random.dirichlet = lambda alpha, size=None: np.full(len(alpha), 1.0 / len(alpha))
    cfg = m.make_cfg(c["sims"])
    cfg.play.search_threads = c["K"]
    pipe = m.stub_net.StubPipe(m.stub_fn(dict(kind="hash", salt=c["salt"])))
    pl = m.ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
    pl.action(c["state"], 0, no_act=c.get("no_act"))
    node = pl.tree[c["state"]]
    n = [int(node.a[mv].n) if mv in node.a else 0 for mv in node.legal_moves]
    pl.close()
    return ci, run, n, int(node.sum_n), pipe.n_positions


def main():
    runs = int(sys.argv_saved[1]) if len(sys.argv_saved) > 1 else 32
    jobs = [(ci, r) for ci in range(len(CASES)) for r in range(runs)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count())) as pool:
        res = pool.map(one, jobs, chunksize=1)
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
              "TV to the mean: median %.3f max %.3f" % (np.median(tv), tv.max()), "sum_n", set(rec["sum_n"]), flush=True)
    with open(os.path.join(HERE, "kgt1_spread.json"), "w") as f:
        json.dump({"meta": m.meta(), "cases": out}, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
