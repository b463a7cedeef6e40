// xq_noise.h -- the root Dirichlet noise generator (host + device).
//
// Reference: select_action_q_and_u redraws  np.random.dirichlet(alpha * ones(n))[0]  for every move at every root visit
// (cchess_alphazero/agent/player.py:304).  NumPy's global generator cannot be matched draw for draw, so the engine
// reproduces the DISTRIBUTION: Dirichlet(alpha 1_n)[0] = X / (X + Y), X ~ Gamma(alpha), Y ~ Gamma(alpha (n - 1)),
// i.e. Beta(alpha, alpha (n - 1)); tests check it against the exact Beta marginal and NumPy's sampler (Kolmogorov-
// Smirnov on 1e5 draws: tests/test_noise_cpu.py on the host build of this file, tests/test_gpu_noise.py on the GPU).
//
// Cost matters: a self-play round draws K x (root moves) values per game (8 x 44 x 4096 = 1.4 M per round), and the draw
// kernel was 22 % of the tree kernels' time with Philox4x32-10 underneath (~25 integer instructions per uniform, 7+
// uniforms per draw).  This version:
//   * uniforms from a counter-based integer hash (NoiseRng: a 3-multiply 32-bit mixer keyed by two words, ~10
//     instructions per uniform) -- statistically plain, not cryptographic; the stream of a draw is addressed by
//     (seed, game, epoch, simulation slot, move), so draws are reproducible and independent of launch geometry;
//   * ONE Box-Muller pair feeds the first Marsaglia-Tsang attempt of BOTH Gamma draws (cos / sin branch), so the common
//     path of a draw is 5 uniforms and 6 transcendental instructions; rejected attempts (5 % / 1 %) draw fresh pairs;
//   * the quotient is formed on the small side (x / s or 1 - y / s) in float32 and widened, which keeps the resolution
//     near 1 that the two-move case needs (Beta(0.2, 0.2) has 1.8 % of its mass within 6e-8 of 1) without a float64
//     division.
// Everything here is XQ_HD so that the CPU test runs the very same code path (libm instead of the hardware
// approximations: same distribution to float32 rounding).
#pragma once
#include <stdint.h>
#include <math.h>
#include "xq_lane.h"

namespace xq {

struct NoiseRng {
    uint32_t a, b, i;
    // key of one draw: TWO independently mixed words (a, a1) from (seed, game, epoch) -- uniform over the wave's lanes, the
    // compiler keeps them on the scalar unit -- and b from (a1, slot, move).  mix() is a bijection, so two (game, epoch)
    // pairs share a stream only when both words collide: a 64-bit key.  (Round 3 derived b from a: ~4e7 (game, epoch) keys
    // per run on a 32-bit word meant ~1e5 pairs of root batches with bit-identical noise rows -- ADVICE r03.)
    static XQ_HD uint32_t mix(uint32_t x)
    {
        x ^= x >> 17; x *= 0xed5ad4bbu;
        x ^= x >> 11; x *= 0xac4c1b51u;
        x ^= x >> 15; x *= 0x31848babu;
        x ^= x >> 14;
        return x;
    }
    static XQ_HD NoiseRng make(uint64_t seed, uint32_t game_key, uint32_t epoch, uint32_t sim, uint32_t move)
    {
        const uint32_t t = mix(game_key * 0x9E3779B1u + (uint32_t)seed);
        const uint32_t a = mix(t ^ (epoch * 0x85EBCA77u + (uint32_t)(seed >> 32)));
        const uint32_t t1 = mix(game_key * 0xB5297A4Du + ((uint32_t)(seed >> 32) ^ 0x68E31DA4u));
        const uint32_t a1 = mix(t1 ^ (epoch * 0x1B56C4E9u + (uint32_t)seed));
        const uint32_t b = mix(a1 + ((sim << 8) | move) * 0xC2B2AE3Du + 0x27D4EB2Fu);
        return NoiseRng{a, b, 0u};
    }
    // the two key words of a (seed, game, epoch) triple (tests: no two triples may share both)
    static XQ_HD uint64_t key64(uint64_t seed, uint32_t game_key, uint32_t epoch)
    {
        const NoiseRng r = make(seed, game_key, epoch, 0u, 0u);
        return ((uint64_t)r.a << 32) | r.b;
    }
    XQ_HD uint32_t bits()
    {
        uint32_t x = a + (i++) * 0x9E3779B9u;
        x ^= x >> 17; x *= 0xed5ad4bbu;
        x ^= b;                              // the second key word enters mid-way: streams with different b share no values
        x ^= x >> 11; x *= 0xac4c1b51u;
        x ^= x >> 15; x *= 0x31848babu;
        x ^= x >> 14;
        return x;
    }
    XQ_HD float next() { return ((float)(bits() >> 8) + 0.5f) * (1.0f / 16777216.0f); }      // (0, 1), 24 bits
};

namespace noise_detail {
#if defined(__HIP_DEVICE_COMPILE__)
// the hardware approximations (1 ulp class): v_log_f32 / v_exp_f32 are base 2, v_cos_f32 / v_sin_f32 take revolutions
XQ_HD float lg2(float x) { return __builtin_amdgcn_logf(x); }
XQ_HD float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
XQ_HD float cos2pi(float u) { return __builtin_amdgcn_cosf(u); }
XQ_HD float sin2pi(float u) { return __builtin_amdgcn_sinf(u); }
XQ_HD float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
XQ_HD float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
XQ_HD float sq(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
XQ_HD float lg2(float x) { return log2f(x); }
XQ_HD float ex2(float x) { return exp2f(x); }
XQ_HD float cos2pi(float u) { return cosf(6.2831853f * u); }
XQ_HD float sin2pi(float u) { return sinf(6.2831853f * u); }
XQ_HD float rsq(float x) { return 1.0f / sqrtf(x); }
XQ_HD float rcp(float x) { return 1.0f / x; }
XQ_HD float sq(float x) { return sqrtf(x); }
#endif
constexpr float LN2 = 0.69314718f;
}  // namespace noise_detail

// Gamma(a, 1) by Marsaglia-Tsang (a < 1: Gamma(a + 1) U^(1/a)).  z: a standard normal for the first attempt.
XQ_HD float gamma_draw(float a, float z, NoiseRng& rng)
{
    using namespace noise_detail;
    float boost = 1.0f;
    if (a <= 0.0f) return 0.0f;
    if (a < 1.0f) {
        boost = ex2(lg2(rng.next()) * rcp(a));                 // U^(1/a)
        a += 1.0f;
    }
    const float d = a - 1.0f / 3.0f, c = rsq(9.0f * d);
    for (int it = 0; it < 32; ++it) {
        float v = 1.0f + c * z;
        if (v > 0.0f) {
            v = v * v * v;
            const float u = rng.next();
            const float z2 = z * z;
            if (u < 1.0f - 0.0331f * z2 * z2) return boost * d * v;
            if (LN2 * lg2(u) < 0.5f * z2 + d * (1.0f - v + LN2 * lg2(v))) return boost * d * v;
        }
        const float u1 = rng.next(), u2 = rng.next();          // rejected: a fresh normal (Box-Muller, cos branch)
        z = sq(-2.0f * LN2 * lg2(u1)) * cos2pi(u2);
    }
    return boost * d;
}

// np.random.dirichlet(alpha * ones(nm))[0]
XQ_HD double dirichlet0(float alpha, int nm, NoiseRng& rng)
{
    using namespace noise_detail;
    const float u1 = rng.next(), u2 = rng.next();
    const float r = sq(-2.0f * LN2 * lg2(u1));
    const float x = gamma_draw(alpha, r * cos2pi(u2), rng);
    const float y = nm > 1 ? gamma_draw(alpha * (float)(nm - 1), r * sin2pi(u2), rng) : 0.0f;
    const float s = x + y;
    if (!(s > 0.0f)) return 1.0 / (double)nm;
    // the quotient on the SMALL side: near 1 the float32 grid (6e-8) is far coarser than the distribution
    const float inv = rcp(s);
    return x >= y ? 1.0 - (double)(y * inv) : (double)(x * inv);
}

}  // namespace xq
