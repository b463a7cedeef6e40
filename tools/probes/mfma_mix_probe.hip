// tools/probes/mfma_mix_probe.hip -- a standalone measurement, NOT part of libczero.so.
//
// Question ("cheaper correction terms"): the residual tower is bound by the socket's power cap, and its
// arithmetic is three bf16 MFMAs per product (w_hi x_hi + w_lo x_hi + w_hi x_lo).  With an fp16 main term (11 bits) the two
// correction terms need only ~4 significant bits, i.e. they could be block-scaled fp8 (or fp6) MFMAs with K = 64.  How much
// faster does the matrix pipe retire that mix IN THE SUSTAINED, POWER-CAPPED STATE?  This program runs register-resident
// MFMA loops (one wave per SIMD, three 32 x 32 accumulator tiles per wave like the tower's K loop, operands changing from
// one MFMA to the next) for a few seconds per mix and prints the time per "K = 64 block" (the work of 64 input channels
// of one tap for three pixel tiles):
//     mode 0   bf16 x3          36 x v_mfma_f32_32x32x16_bf16                                  (today's arithmetic)
//     mode 1   f16 + 2 x fp8    12 x v_mfma_f32_32x32x16_f16 + 6 x v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)
//     mode 2   bf16 x1          12 x v_mfma_f32_32x32x16_bf16                                  (plain bf16, for scale)
//     mode 3   fp8 only          6 x v_mfma_scale_f32_32x32x64_f8f6f4
//     mode 4   f16 + 2 x fp6    12 x f16 + 6 x f8f6f4 with cbsz = blgp = 2 (e2m3)
// It also checks what the kernel would rely on: the scaled fp8 MFMA's value semantics (e4m3 operands, E8M0 block scales,
// the 32 x 32 accumulator layout) against a host computation.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_mix_probe.hip -o tools/probes/mfma_mix_probe
//     tools/probes/mfma_mix_probe <mode> <seconds>         |      tools/probes/mfma_mix_probe check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
    return x;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float)(int32_t)h * (1.0f / 2147483648.0f); }   // [-1, 1)

template <typename V8, typename E> __device__ __forceinline__ V8 rand_frag(uint32_t key)
{
    V8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (E)unit(mix32(key * 8u + j));
    return v;
}
__device__ __forceinline__ i32x8 rand_f8(uint32_t key, uint32_t mask)
{
    i32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (int)(mix32(key * 8u + j) & mask);
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_mix(float* out, long iters, int scale)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    if (MODE == 0 || MODE == 2) {
        bf16x8 ah[4], al[4], bh[3][4], bl[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ah[k] = rand_frag<bf16x8, __bf16>(tid * 64 + k);
            al[k] = rand_frag<bf16x8, __bf16>(tid * 64 + 4 + k);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bh[t][k] = rand_frag<bf16x8, __bf16>(tid * 64 + 8 + t * 4 + k);
                bl[t][k] = rand_frag<bf16x8, __bf16>(tid * 64 + 20 + t * 4 + k);
            }
        }
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[k], bh[t][k], acc[t], 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[k], bh[t][k], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[k], bl[t][k], acc[t], 0, 0, 0);
                }
            }
        }
    } else {
        constexpr int FMT = MODE == 4 ? 2 : 0;                    // 0: fp8 e4m3, 2: fp6 e2m3
        // random operand bytes; e4m3: clear the top exponent bit (|x| < 2, no NaN); fp6 words: any bit pattern is a number
        const uint32_t mask = MODE == 4 ? 0xFFFFFFFFu : 0xBFBFBFBFu;
        f16x8 ah[4], bh[3][4];
        i32x8 a8[2], b8[3][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ah[k] = rand_frag<f16x8, _Float16>(tid * 64 + k);
#pragma unroll
            for (int t = 0; t < 3; ++t) bh[t][k] = rand_frag<f16x8, _Float16>(tid * 64 + 8 + t * 4 + k);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a8[c] = rand_f8(tid * 64 + 32 + c, mask);
#pragma unroll
            for (int t = 0; t < 3; ++t) b8[t][c] = rand_f8(tid * 64 + 40 + t * 2 + c, mask);
        }
        for (long it = 0; it < iters; ++it) {
            if (MODE != 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k], bh[t][k], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[c], b8[t][c], acc[t], FMT, FMT, 0, scale, 0, scale);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[tid] = s;
}

// ---- value semantics of the scaled fp8 MFMA ---------------------------------------------------------------------------
__global__ void k_check(const uint8_t* a, const uint8_t* b, float* c, int scale_a, int scale_b)
{
    // a: [32 rows][64 k] e4m3 bytes, b: [32 cols][64 k]; lane l holds row / column l % 32, k = (l / 32) * 32 .. + 31
    const int l = threadIdx.x;
    i32x8 va, vb;
    memcpy(&va, a + (l % 32) * 64 + (l / 32) * 32, 32);
    memcpy(&vb, b + (l % 32) * 64 + (l / 32) * 32, 32);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 0, 0, 0, scale_a, 0, scale_b);
#pragma unroll
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];   // [row of a][row of b]
}

static float e4m3_to_float(uint8_t v)
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

static uint8_t float_to_e4m3(float x)          // exact inputs only (test patterns)
{
    for (int v = 0; v < 256; ++v)
        if ((v & 0x7F) != 0x7F && e4m3_to_float((uint8_t)v) == x && !(v == 0x80)) return (uint8_t)v;
    return 0;
}

static int run_check()
{
    std::vector<uint8_t> a(32 * 64), b(32 * 64);
    uint8_t *da, *db;
    float* dc;
    CK(hipMalloc(&da, a.size())); CK(hipMalloc(&db, b.size())); CK(hipMalloc(&dc, 32 * 32 * 4));
    std::vector<float> c(32 * 32);
    auto launch = [&](int sa, int sb) {
        CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, da, db, dc, sa, sb);
        CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
    };
    int bad = 0;
    const uint8_t one = float_to_e4m3(1.0f);
    // 1. all ones: 64 x 2^(sa - 127) x 2^(sb - 127)
    for (auto& v : a) v = one;
    for (auto& v : b) v = one;
    const int scales[4][2] = {{127, 127}, {115, 127}, {127, 121}, {120, 130}};
    for (auto& sc : scales) {
        launch(sc[0], sc[1]);
        const float want = ldexpf(64.0f, sc[0] - 127 + sc[1] - 127);
        int wrong = 0;
        for (float v : c) wrong += v != want;
        printf("check ones   scale_a=%d scale_b=%d  c[0]=%g want %g  wrong entries %d  %s\n", sc[0], sc[1], c[0], want, wrong,
               wrong ? "MISMATCH" : "OK");
        bad += wrong != 0;
    }
    // 2. one-hot rows: a[i][k] = [k == (5 i + 3) % 64], b[j][k] = 2^(j % 4) [k == (7 j + 1) % 64]: c[i][j] = 2^(j % 4) when the two
    //    k agree -- checks that the A and B operands use the same lane -> k map and the accumulator layout assumed above
    memset(a.data(), 0, a.size()); memset(b.data(), 0, b.size());
    for (int i = 0; i < 32; ++i) a[i * 64 + (5 * i + 3) % 64] = one;
    for (int j = 0; j < 32; ++j) b[j * 64 + (7 * j + 1) % 64] = float_to_e4m3(ldexpf(1.0f, j % 4));
    launch(127, 127);
    {
        int wrong = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float want = (5 * i + 3) % 64 == (7 * j + 1) % 64 ? ldexpf(1.0f, j % 4) : 0.0f;
                wrong += c[i * 32 + j] != want;
            }
        printf("check onehot wrong entries %d  %s\n", wrong, wrong ? "MISMATCH" : "OK");
        bad += wrong != 0;
    }
    // 3. random operands: error relative to the sum of |terms| (what the accumulation can be held to)
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (auto& v : a) v = (uint8_t)(rnd() & 0xBF);
    for (auto& v : b) v = (uint8_t)(rnd() & 0xBF);
    launch(127, 127);
    {
        double worst = 0.0, worst_rel = 0.0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0.0, mag = 0.0;
                for (int k = 0; k < 64; ++k) {
                    const double t = (double)e4m3_to_float(a[i * 64 + k]) * e4m3_to_float(b[j * 64 + k]);
                    ref += t; mag += fabs(t);
                }
                const double d = fabs(ref - c[i * 32 + j]);
                if (d / mag > worst) worst = d / mag;
                if (d / (fabs(ref) + 1e-30) > worst_rel) worst_rel = d / (fabs(ref) + 1e-30);
            }
        printf("check random max |diff| / sum|terms| %.3g   max |diff| / |result| %.3g   %s\n", worst, worst_rel,
               worst < 1e-6 ? "OK" : "MISMATCH");
        printf("       c[0][0..3] = %g %g %g %g\n", c[0], c[1], c[2], c[3]);
        bad += worst >= 1e-6;
    }
    return bad;
}

template <int MODE> static double run_mode(long iters, float* out, int blocks)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 127);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && !strcmp(argv[1], "check")) return run_check();
    const int mode = argc >= 2 ? atoi(argv[1]) : 0;
    const double seconds = argc >= 3 ? atof(argv[2]) : 3.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    float* out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    auto run = [&](long iters) {
        switch (mode) {
        case 0: return run_mode<0>(iters, out, blocks);
        case 1: return run_mode<1>(iters, out, blocks);
        case 2: return run_mode<2>(iters, out, blocks);
        case 3: return run_mode<3>(iters, out, blocks);
        default: return run_mode<4>(iters, out, blocks);
        }
    };
    const long probe = 20000;
    run(probe);                                           // code load
    const double ms_probe = run(probe);                   // cold-clock estimate of the rate
    // the timed run is cut into ~0.5 s launches so that the rate can be seen settling under the power cap
    const long chunk = (long)(probe * 500.0 / ms_probe);
    const int n = (int)(seconds / 0.5) + 1;
    printf("mode %d: %d CUs x 4 waves, %ld blocks of K=64 per launch\n", mode, blocks, chunk);
    double last = 0.0, first = 0.0;
    for (int i = 0; i < n; ++i) {
        const double ms = run(chunk);
        if (i == 0) first = ms;
        last = ms;
        printf("  launch %2d  %.1f ms  %.2f ns per K=64 block per wave\n", i, ms, ms * 1e6 / chunk);
    }
    printf("RESULT mode %d ns_per_block_first %.3f ns_per_block_settled %.3f\n", mode, first * 1e6 / chunk, last * 1e6 / chunk);
    return 0;
}
