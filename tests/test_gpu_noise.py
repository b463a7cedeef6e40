"""Root Dirichlet noise (agent/player.py:304: np.random.dirichlet(alpha * ones(n))[0], redrawn per move per root
visit) -- the one part of the search that can only match the reference in DISTRIBUTION (NumPy's global RNG cannot be
reproduced on the device).  SURVEY section 7: "validated statistically".

  * the generator of k_noise (float32 Marsaglia-Tsang Gamma draws on a Philox4x32-10 stream) against the exact
    marginal Beta(alpha, alpha (n - 1)) and against NumPy's own sampler: Kolmogorov-Smirnov on 10^5 draws for
    n in {2, 20, 44, 68} x alpha in {0.2, 0.3};
  * the effect on the search: entropy / support / top share of the root visit distribution over 256 seeded searches of
    the reference player (tests/golden/noise_ref.json, recorded by make_golden_noise.py with np.random.seed) against
    512 searches of the engine with different noise seeds.
"""
import json
import os

import numpy as np
import pytest

import stub_net
from oracle import xq_oracle as xo
from test_gpu_search import boards_tensor, gpu, play_config, stub_eval  # noqa: F401  (gpu: the fixture)

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

N_DRAWS = 100000
KS_CRITICAL = 1.95 / np.sqrt(N_DRAWS)        # p = 0.001 one-sample; 16 cases -> ~1.6 % family-wise false alarm


@pytest.mark.parametrize("alpha", [0.2, 0.3])
@pytest.mark.parametrize("n_moves", [2, 20, 44, 68])
def test_noise_marginal_matches_numpy_dirichlet(gpu, alpha, n_moves):
    from scipy import stats
    x = gpu.S.debug_noise(alpha, n_moves, N_DRAWS, seed=20260924, game_key=n_moves).cpu().numpy()
    assert x.shape == (N_DRAWS,) and np.isfinite(x).all() and (x >= 0).all() and (x <= 1).all()
    beta = stats.beta(alpha, alpha * (n_moves - 1))
    d_exact = stats.kstest(x, beta.cdf).statistic
    assert d_exact < KS_CRITICAL, (alpha, n_moves, d_exact, KS_CRITICAL)
    ref = np.random.default_rng(7).dirichlet(alpha * np.ones(n_moves), N_DRAWS)[:, 0]      # NumPy's own sampler
    d_np = stats.ks_2samp(x, ref).statistic
    assert d_np < 1.95 * np.sqrt(2.0 / N_DRAWS), (alpha, n_moves, d_np)
    # first two moments of Beta(a, a (n - 1)): mean 1/n, variance (n - 1) / (n^2 (a n + 1))
    se_mean = np.sqrt(beta.var() / N_DRAWS)
    assert abs(x.mean() - 1.0 / n_moves) < 5 * se_mean
    assert abs(x.var() - beta.var()) < 0.03 * beta.var() + 1e-6


def test_draws_differ_across_streams_and_repeat_for_one(gpu):
    a = gpu.S.debug_noise(0.2, 44, 4096, seed=1, game_key=5).cpu().numpy()
    b = gpu.S.debug_noise(0.2, 44, 4096, seed=1, game_key=5).cpu().numpy()
    c = gpu.S.debug_noise(0.2, 44, 4096, seed=1, game_key=6).cpu().numpy()
    d = gpu.S.debug_noise(0.2, 44, 4096, seed=2, game_key=5).cpu().numpy()
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)
    assert abs(np.corrcoef(a, c)[0, 1]) < 0.08 and abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.08


def _visit_stats(n):
    n = np.asarray(n, dtype=np.float64)
    p = n / n.sum()
    nz = p[p > 0]
    return float(-(nz * np.log(nz)).sum()), int((n > 0).sum()), float(p.max())


def test_root_visit_distribution_matches_the_reference_with_noise(gpu):
    """256 reference searches (K = 1, noise_eps 0.25, alpha 0.2, np.random.seed(i)) vs 512 engine searches with
    different seeds, same stub network: the noise changes which root moves get visited, so entropy / support / top
    share of the visit counts are compared as populations (Welch z-score of the means, KS of the entropies)."""
    from scipy import stats
    path = os.path.join(HERE, "golden", "noise_ref.json")
    if not os.path.exists(path):
        pytest.skip("noise_ref.json not generated")
    with open(path) as f:
        ref = json.load(f)
    G = 512
    for case in ref["cases"]:
        pc = play_config(simulation_num_per_move=case["sims"], search_threads=1, noise_eps=case["noise_eps"],
                         dirichlet_alpha=case["alpha"], c_puct=case["c_puct"])
        s = gpu.S.Search(pc, G, seed=991)
        s.set_roots(boards_tensor(gpu, [case["state"]] * G))
        s.run_until_idle(stub_eval(gpu, case["stub"]))
        st = s.root_stats()
        s.close()
        got = np.array([_visit_stats(st["n"][g, :int(st["counts"][g])]) for g in range(G)])
        exp = np.array([_visit_stats(r) for r in case["visits"]])
        assert all(int(st["n"][g].sum()) == case["sims"] - 1 for g in range(G))
        for k, name in enumerate(("entropy", "support", "top_share")):
            a, b = got[:, k], exp[:, k]
            z = (a.mean() - b.mean()) / np.sqrt(a.var(ddof=1) / len(a) + b.var(ddof=1) / len(b) + 1e-30)
            assert abs(z) < 4.0, (case["name"], name, a.mean(), b.mean(), z)
        ks = stats.ks_2samp(got[:, 0], exp[:, 0])
        assert ks.pvalue > 1e-3, (case["name"], ks)
        # and the noise matters: the same searches without noise all give ONE visit vector
        e0 = _visit_stats(case["visits_no_noise"])[0]
        assert exp[:, 0].std() > 0 and got[:, 0].std() > 0 and abs(got[:, 0].mean() - e0) > 0 or case["noise_eps"] == 0
