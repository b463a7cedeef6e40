"""Reference side of tests/test_gpu_noise.py: root visit counts of the REFERENCE player with Dirichlet noise on
(agent/player.py:304), one K = 1 search per seed (np.random.seed(i) -- the reference draws the noise from NumPy's global
RNG), hash-stub network.  Output: noise_ref.json (visit vectors in legal-move order).

    python tests/golden/make_golden_noise.py [n_seeds]
"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv_saved, sys.argv = sys.argv, [sys.argv[0], "none"]
sys.path.insert(0, HERE)
import make_golden_mcts as m  # noqa: E402  (sets up the reference imports, the stub network, thread switching)
import numpy as np  # noqa: E402

CASES = [
    dict(name="init_mini", state=m.senv.INIT_STATE, sims=100, noise_eps=0.25, alpha=0.2, c_puct=1.5,
         stub=dict(kind="hash", salt=71)),
    dict(name="mid_normal", state='r1e1s1e1r/4m4/2k1c1k2/p1p1p1p1p/9/2P6/P3P1P1P/1CK1C1K2/9/R1EMSME1R', sims=160,
         noise_eps=0.15, alpha=0.2, c_puct=1.5, stub=dict(kind="hash", salt=72)),
    dict(name="init_alpha03", state=m.senv.INIT_STATE, sims=120, noise_eps=0.4, alpha=0.3, c_puct=3.0,
         stub=dict(kind="uniform", value=0.0)),
]


def one(job):
    ci, seed = job
    c = CASES[ci]
    cfg = m.make_cfg(c["sims"], c_puct=c["c_puct"])
    cfg.play.noise_eps = c["noise_eps"] if seed >= 0 else 0
    cfg.play.dirichlet_alpha = c["alpha"]
    np.random.seed(max(seed, 0))
    pipe = m.stub_net.StubPipe(m.stub_fn(c["stub"]))
    pl = m.ref_player.CChessPlayer(cfg, search_tree=None, pipes=pipe, enable_resign=False)
    pl.action(c["state"], 0)
    node = pl.tree[c["state"]]
    n = [int(node.a[mv].n) if mv in node.a else 0 for mv in node.legal_moves]
    pl.close()
    return ci, seed, n


def main():
    n_seeds = int(sys.argv_saved[1]) if len(sys.argv_saved) > 1 else 256
    jobs = [(ci, s) for ci in range(len(CASES)) for s in range(-1, n_seeds)]
    with mp.get_context("fork").Pool(min(8, os.cpu_count())) as pool:
        res = pool.map(one, jobs, chunksize=8)
    out = []
    for ci, c in enumerate(CASES):
        rec = dict(c)
        rec["visits"] = [n for (i, s, n) in res if i == ci and s >= 0]
        rec["visits_no_noise"] = next(n for (i, s, n) in res if i == ci and s < 0)
        out.append(rec)
        ent = [-(np.array(v) / sum(v) * np.log(np.maximum(np.array(v) / sum(v), 1e-300))).sum() for v in rec["visits"]]
        print(c["name"], "seeds", len(rec["visits"]), "mean entropy %.4f sd %.4f" % (np.mean(ent), np.std(ent)))
    with open(os.path.join(HERE, "noise_ref.json"), "w") as f:
        json.dump({"meta": m.meta(), "cases": out}, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
