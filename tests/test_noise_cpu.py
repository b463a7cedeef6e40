"""CPU check of the root Dirichlet noise generator (csrc/xq_noise.h compiled with g++): the distribution the reference
draws per move per root visit -- np.random.dirichlet(alpha * ones(n))[0], agent/player.py:304 -- i.e. Beta(alpha,
alpha (n - 1)).  Same criteria as the GPU test (tests/test_gpu_noise.py): Kolmogorov-Smirnov against the exact marginal
and against NumPy's own sampler on 1e5 draws, first two moments; plus the integer hash underneath (uniformity, serial and
cross-stream correlation).  The GPU runs the same code with the hardware log / exp / sin / cos / rcp approximations."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DRAWS = 100000
KS_CRITICAL = 1.95 / np.sqrt(N_DRAWS)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("noise") / "libnoise.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "noise_harness.cpp"), "-o", str(out)])
    L = C.CDLL(str(out))
    L.noise_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_void_p, C.c_int]
    L.noise_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
    L.noise_first_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    L.noise_uniforms_used.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_double)]
    return L


def draws(L, alpha, nm, n, seed=20260924, key=None):
    out = np.zeros(n, dtype=np.float64)
    L.noise_draws(seed, nm if key is None else key, alpha, nm, out.ctypes.data_as(C.c_void_p), n)
    return out


@pytest.mark.parametrize("alpha", [0.2, 0.3, 1.5])
@pytest.mark.parametrize("n_moves", [1, 2, 3, 20, 44, 68, 128])
def test_marginal_is_beta(harness, alpha, n_moves):
    from scipy import stats
    x = draws(harness, alpha, n_moves, N_DRAWS)
    assert np.isfinite(x).all() and (x >= 0).all() and (x <= 1).all()
    if n_moves == 1:
        assert (x == 1.0).all()                             # dirichlet of one component
        return
    beta = stats.beta(alpha, alpha * (n_moves - 1))
    d = stats.kstest(x, beta.cdf).statistic
    assert d < KS_CRITICAL, (alpha, n_moves, d, KS_CRITICAL)
    ref = np.random.default_rng(7).dirichlet(alpha * np.ones(n_moves), N_DRAWS)[:, 0]
    assert stats.ks_2samp(x, ref).statistic < 1.95 * np.sqrt(2.0 / N_DRAWS)
    assert abs(x.mean() - 1.0 / n_moves) < 5 * np.sqrt(beta.var() / N_DRAWS)
    assert abs(x.var() - beta.var()) < 0.03 * beta.var() + 1e-6


def test_two_move_case_keeps_its_resolution_near_one(harness):
    """Beta(0.2, 0.2) puts 1.8 % of its mass within 6e-8 of 1: a float32 quotient would collapse it onto exactly 1.0
    (the defect the KS test found in round 1); the small-side quotient must not."""
    x = draws(harness, 0.2, 2, N_DRAWS)
    # (exactly 1.0 only where float64 itself runs out: P(1 - X < 1.1e-16) = 3.4e-4 for Beta(0.2, 0.2), NumPy's too)
    assert (x == 1.0).mean() < 8e-4 and (x == 0.0).mean() < 2e-4
    near = ((x > 1 - 6e-8) & (x < 1)).mean()
    assert 0.01 < near < 0.03, near


def test_streams_repeat_and_differ(harness):
    a, b = draws(harness, 0.2, 44, 4096, seed=1, key=5), draws(harness, 0.2, 44, 4096, seed=1, key=5)
    c, d = draws(harness, 0.2, 44, 4096, seed=1, key=6), draws(harness, 0.2, 44, 4096, seed=2, key=5)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)
    assert abs(np.corrcoef(a, c)[0, 1]) < 0.08 and abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.08


def test_the_integer_hash_is_uniform_and_uncorrelated(harness):
    from scipy import stats
    n = 1 << 18
    u = np.zeros(n, dtype=np.float32)
    harness.noise_uniforms(123, 7, 3, 2, 11, u.ctypes.data_as(C.c_void_p), n)
    assert 0.0 < u.min() and u.max() < 1.0
    assert stats.kstest(u.astype(np.float64), "uniform").statistic < 1.95 / np.sqrt(n)
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 4 / np.sqrt(n)
    assert abs(np.corrcoef(u[:-2], u[2:])[0, 1]) < 4 / np.sqrt(n)
    chi = stats.chisquare(np.bincount((u * 256).astype(int), minlength=256)).pvalue
    assert chi > 1e-4
    # first uniforms of neighbouring streams (sim x move grid of one game and epoch, as the lanes of k_noise start)
    g = np.zeros(64 * 128, dtype=np.float32)
    harness.noise_first_uniforms(123, 7, 3, g.ctypes.data_as(C.c_void_p), 64, 128)
    assert stats.kstest(g.astype(np.float64), "uniform").statistic < 1.95 / np.sqrt(g.size)
    m = g.reshape(64, 128)
    assert abs(np.corrcoef(m[:, :-1].ravel(), m[:, 1:].ravel())[0, 1]) < 0.05
    assert abs(np.corrcoef(m[:-1].ravel(), m[1:].ravel())[0, 1]) < 0.05
    assert len(np.unique(g)) > 0.99 * g.size


def test_cost_in_uniforms(harness):
    """the common path of a draw is 5 uniforms (one Box-Muller pair shared by both Gamma draws)"""
    mean = C.c_double()
    worst = harness.noise_uniforms_used(9, 9, 0.2, 44, 20000, C.byref(mean))
    assert 5.0 <= mean.value < 5.6 and worst < 40, (mean.value, worst)


def test_stream_keys_are_64_bit(harness):
    """a self-play run addresses ~4e7 (game, epoch) root batches; with a 32-bit key ~1e5 pairs of them would draw
    bit-identical noise rows (birthday bound).  The key is two independently mixed words: over 2e6 (game, epoch) pairs of a
    4096-game run the first word alone DOES collide, the pair never."""
    n = 2_000_000
    keys = np.zeros(n, dtype=np.uint64)
    harness.noise_keys.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_int]
    harness.noise_keys(20260923, 4096, keys.ctypes.data_as(C.c_void_p), n)
    assert len(np.unique(keys)) == n
    first_word = (keys >> np.uint64(32)).astype(np.uint32)
    assert len(np.unique(first_word)) < n                    # ~470 expected collisions at 2e6 keys on 32 bits
    # and two triples that differ only in the seed's high word, or only in the game, get different streams
    a = np.zeros(8, dtype=np.float32)
    b = np.zeros(8, dtype=np.float32)
    harness.noise_uniforms(1, 7, 3, 2, 11, a.ctypes.data_as(C.c_void_p), 8)
    harness.noise_uniforms(1 + (1 << 32), 7, 3, 2, 11, b.ctypes.data_as(C.c_void_p), 8)
    assert not np.array_equal(a, b)
