// tests/noise_harness.cpp -- CPU build of the root-noise generator (chinesechess-alphazero_amd/csrc/xq_noise.h), the
// same code k_noise runs on the GPU with libm in place of the hardware approximations.  Built with g++ by
// tests/test_noise_cpu.py; the stream addressing mirrors k_debug_noise (csrc/xq_search.hip).
#include <stdint.h>
#include "../chinesechess-alphazero_amd/csrc/xq_noise.h"

using namespace xq;

extern "C" {

void noise_draws(uint64_t seed, uint32_t game_key, double alpha, int nm, double* out, int n)
{
    for (int i = 0; i < n; ++i) {
        const uint32_t epoch = (uint32_t)(i / (8 * 128)), sim = (uint32_t)(i / 128) % 8u, j = (uint32_t)i % 128u;
        NoiseRng rng = NoiseRng::make(seed, game_key, epoch, sim, j);
        out[i] = dirichlet0((float)alpha, nm, rng);
    }
}

// raw uniforms of one stream (tests of the integer hash itself)
void noise_uniforms(uint64_t seed, uint32_t game_key, uint32_t epoch, uint32_t sim, uint32_t move, float* out, int n)
{
    NoiseRng rng = NoiseRng::make(seed, game_key, epoch, sim, move);
    for (int i = 0; i < n; ++i) out[i] = rng.next();
}

// the first uniform of many streams (what neighbouring lanes of k_noise draw first)
void noise_first_uniforms(uint64_t seed, uint32_t game_key, uint32_t epoch, float* out, int n_sims, int n_moves)
{
    for (int s = 0; s < n_sims; ++s)
        for (int j = 0; j < n_moves; ++j) {
            NoiseRng rng = NoiseRng::make(seed, game_key, epoch, (uint32_t)s, (uint32_t)j);
            out[s * n_moves + j] = rng.next();
        }
}

// the 64-bit stream key of n (game, epoch) pairs: game = i % n_games, epoch = i / n_games
void noise_keys(uint64_t seed, int n_games, uint64_t* out, int n)
{
    for (int i = 0; i < n; ++i) out[i] = NoiseRng::key64(seed, (uint32_t)(i % n_games), (uint32_t)(i / n_games));
}

int noise_uniforms_used(uint64_t seed, uint32_t game_key, double alpha, int nm, int n, double* mean_out)
{
    long long used = 0;
    int worst = 0;
    for (int i = 0; i < n; ++i) {
        NoiseRng rng = NoiseRng::make(seed, game_key, (uint32_t)i, 0u, 0u);
        (void)dirichlet0((float)alpha, nm, rng);
        used += rng.i;
        if ((int)rng.i > worst) worst = (int)rng.i;
    }
    *mean_out = (double)used / n;
    return worst;
}

}
