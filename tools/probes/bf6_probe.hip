// tools/probes/bf6_probe.hip -- a standalone check, NOT part of libczero.so: the semantics the c6 arithmetic relies on.
//   1. v_cvt_scalef32_2xpk16_bf6_f32: element order of the 32 packed values, rounding, saturation, the scale operand
//   2. v_cvt_scalef32_pk32_f32_bf6: the inverse (order, scale)
//   3. v_permlane32_swap
//   4. v_mfma_scale_f32_32x32x64_f8f6f4 with bf6 operands: lane -> (row, k) map shared by A and B, element j of a lane at
//      bits 6 j, the E8M0 scales, and its issue rate against the e4m3 form
//     hipcc --offload-arch=gfx950 -O3 tools/probes/bf6_probe.hip -o tools/probes/bf6_probe && tools/probes/bf6_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(6))) unsigned int u32x6;
typedef __attribute__((ext_vector_type(8))) int i32x8;

static float bf6_value(int code)            // e3m2, bias 3, no inf / nan
{
    const int s = code >> 5, e = (code >> 2) & 7, m = code & 3;
    const float v = e ? ldexpf(1.0f + m / 4.0f, e - 3) : ldexpf(m / 4.0f, -2);
    return s ? -v : v;
}
static int bf6_encode(float x)              // round to nearest even, saturating
{
    const int s = x < 0 || (x == 0 && signbit(x));
    float a = fabsf(x);
    if (a > 28.0f) a = 28.0f;
    int best = 0; float bd = 1e30f;
    for (int c = 0; c < 32; ++c) {
        const float d = fabsf(bf6_value(c) - a);
        if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
    }
    return (s << 5) | best;
}

__global__ void k_cvt(const float* in, float scale, unsigned int* packed, float* back, float back_scale)
{
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    const u32x6 p = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) packed[threadIdx.x * 6 + i] = p[i];
    const f32x32 u = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(p, back_scale);
    for (int i = 0; i < 32; ++i) back[threadIdx.x * 32 + i] = u[i];
}

__global__ void k_swap(int* out)
{
    int a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}

// one wave: D = A(32 x 64) B(64 x 32) with the lane's eight registers as given
template <int FMT>
__global__ void k_mfma(const int* a, const int* b, float* d, int sa, int sb)
{
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[threadIdx.x * 8 + i]; vb[i] = b[threadIdx.x * 8 + i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, FMT, FMT, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) d[threadIdx.x * 16 + i] = acc[i];
}

template <int FMT>
__global__ __launch_bounds__(256) void k_rate(const int* a, float* out, int iters, long long* cyc)
{
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[(threadIdx.x & 63) * 8 + i]; vb[i] = a[(63 - (threadIdx.x & 63)) * 8 + i]; }
    f32x16 acc[3];
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 16; ++i) acc[p][i] = 0.0f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[p], FMT, FMT, 0, 127, 0, 127);
    }
    const long long t1 = clock64();
    float s = 0.0f;
    for (int p = 0; p < 3; ++p) for (int i = 0; i < 16; ++i) s += acc[p][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    // ---- 1, 2: conversions
    std::vector<float> in(64 * 32);
    uint32_t st = 5;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            float v;
            if (l == 0) v = (float)(i + 1) * 0.25f;                      // distinct, mostly exact
            else if (l == 1) v = bf6_value(i) * 8.0f;                     // every positive code, times the scale
            else if (l == 2) v = (i & 1 ? -1.0f : 1.0f) * (24.0f + i);    // saturation above 28 * scale / 8
            else if (l == 3) v = 8.0f * (bf6_value(i % 31) + bf6_value(i % 31 + 1)) * 0.5f;     // ties
            else v = ((int)(rnd() & 0xFFFF) - 32768) / 1024.0f * 8.0f;
            in[l * 32 + i] = v;
        }
    float *din, *dback; unsigned int* dpk;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dback, in.size() * 4)); CK(hipMalloc(&dpk, 64 * 6 * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, 8.0f, dpk, dback, 8.0f);
    std::vector<unsigned int> pk(64 * 6); std::vector<float> back(64 * 32);
    CK(hipMemcpy(pk.data(), dpk, pk.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(back.data(), dback, back.size() * 4, hipMemcpyDeviceToHost));
    int bad_order = 0, bad_round = 0, bad_back = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
            const int bit = 6 * i;
            const uint64_t w = (uint64_t)pk[l * 6 + bit / 32] | ((uint64_t)(bit / 32 + 1 < 6 ? pk[l * 6 + bit / 32 + 1] : 0) << 32);
            const int code = (int)((w >> (bit % 32)) & 63);
            const int want = bf6_encode(in[l * 32 + i] / 8.0f);          // hypothesis: element i at bits 6 i, value / scale, RNE, saturating
            if (code != want) { if (l == 0) ++bad_order; else ++bad_round; if (bad_order + bad_round <= 12) printf("  cvt lane %d elem %d in %g -> code %d (%g), expected %d (%g)\n", l, i, in[l * 32 + i], code, bf6_value(code), want, bf6_value(want)); }
            if (back[l * 32 + i] != bf6_value(code) * 8.0f) { ++bad_back; if (bad_back <= 6) printf("  unpack lane %d elem %d code %d -> %g, expected %g\n", l, i, code, back[l * 32 + i], bf6_value(code) * 8.0f); }
        }
    printf("RESULT cvt 2xpk16 bf6: element i (src0 = 0..15, src1 = 16..31) at bits 6 i, value / scale, RNE, saturating: %s (%d order, %d rounding mismatches); pk32 f32<-bf6 times scale: %s\n",
           bad_order + bad_round ? "NO" : "yes", bad_order, bad_round, bad_back ? "NO" : "yes");
    printf("  lane 0 raw: %08x %08x %08x %08x %08x %08x\n", pk[0], pk[1], pk[2], pk[3], pk[4], pk[5]);
    // ---- 3: permlane32_swap
    int* dsw; CK(hipMalloc(&dsw, 128 * 4));
    hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, dsw);
    std::vector<int> sw(128);
    CK(hipMemcpy(sw.data(), dsw, 512, hipMemcpyDeviceToHost));
    // hypothesis: r[0] = lanes < 32: own a, lanes >= 32: b of lane - 32;  r[1] = lanes < 32: a of lane + 32, lanes >= 32: own b
    int bad_sw = 0;
    for (int l = 0; l < 64; ++l) {
        const int w0 = l < 32 ? 1000 + l : 2000 + l - 32, w1 = l < 32 ? 1000 + l + 32 : 2000 + l;
        if (sw[2 * l] != w0 || sw[2 * l + 1] != w1) ++bad_sw;
    }
    printf("RESULT permlane32_swap(a, b): upper half of a <-> lower half of b: %s   (lane 0: %d %d, lane 40: %d %d)\n", bad_sw ? "NO" : "yes", sw[0], sw[1], sw[80], sw[81]);
    // ---- 4: the bf6 MFMA
    std::vector<int> A(64 * 8, 0), B(64 * 8, 0);
    std::vector<int> ca(64 * 32), cb(64 * 32);
    auto put = [&](std::vector<int>& v, int lane, int j, int code) {
        const int bit = 6 * j;
        uint64_t w = (uint64_t)(unsigned)v[lane * 8 + bit / 32] | ((uint64_t)(unsigned)v[lane * 8 + bit / 32 + 1] << 32);
        w |= (uint64_t)code << (bit % 32);
        v[lane * 8 + bit / 32] = (int)(uint32_t)w; v[lane * 8 + bit / 32 + 1] = (int)(uint32_t)(w >> 32);
    };
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            ca[l * 32 + j] = (int)(rnd() % 64); cb[l * 32 + j] = (int)(rnd() % 64);
            put(A, l, j, ca[l * 32 + j]); put(B, l, j, cb[l * 32 + j]);
        }
    for (int l = 0; l < 64; ++l) { A[l * 8 + 6] = A[l * 8 + 7] = B[l * 8 + 6] = B[l * 8 + 7] = (int)0xDEADBEEF; }    // must be ignored
    int *dA, *dB; float* dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 64 * 16 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_mfma<3>), dim3(1), dim3(64), 0, 0, dA, dB, dD, 127 + 2, 127 - 5);
    std::vector<float> D(64 * 16);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad_mm = 0; double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 32; ++j) ref += (double)bf6_value(ca[(h * 32 + row) * 32 + j]) * bf6_value(cb[(h * 32 + col) * 32 + j]);
            ref *= ldexp(1.0, 2 - 5);
            const double e = fabs(D[l * 16 + r] - ref);
            if (e > worst) worst = e;
            if (e > 1e-3 * (1 + fabs(ref))) { ++bad_mm; if (bad_mm <= 4) printf("  mfma lane %d reg %d: %g, expected %g\n", l, r, D[l * 16 + r], ref); }
        }
    printf("RESULT mfma bf6: lane (h, row) holds k = 32 h + j at bits 6 j for A and for B, scales 2^(byte - 127), registers 6, 7 ignored: %s (worst |d| %.3g)\n",
           bad_mm ? "NO" : "yes", worst);
    // ---- rate
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float* dout; long long* dcyc;
    CK(hipMalloc(&dout, (size_t)prop.multiProcessorCount * 256 * 4)); CK(hipMalloc(&dcyc, 8));
    for (int fmt = 0; fmt < 2; ++fmt) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            if (fmt == 0) hipLaunchKernelGGL((k_rate<0>), dim3(prop.multiProcessorCount), dim3(256), 0, 0, dA, dout, iters, dcyc);
            else hipLaunchKernelGGL((k_rate<3>), dim3(prop.multiProcessorCount), dim3(256), 0, 0, dA, dout, iters, dcyc);
            CK(hipDeviceSynchronize());
        }
        long long cyc = 0;
        CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
        printf("RESULT rate %s: %.1f shader cycles per 32x32x64 MFMA (one wave per SIMD, all CUs)\n", fmt ? "bf6" : "e4m3", (double)cyc / (iters * 24.0));
    }
    return 0;
}
