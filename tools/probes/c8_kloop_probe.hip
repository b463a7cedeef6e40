// tools/probes/c8_kloop_probe.hip -- a standalone measurement, NOT part of libczero.so.
//
// Question: the c8 residual block is co-limited by the LDS port -- per tap and pixel tile a wave reads
// 256 bytes per lane from LDS (fp16 fragments + the two e4m3 pieces) for 8 x 32 + 4 x 64 = 512 cycles of matrix work, and
// four waves ask the port for exactly as many cycles as the matrix pipes.  Giving a wave TWO channel tiles (every pixel
// fragment feeds two MFMAs) halves the LDS reads per MFMA but needs a restructured kernel (two boards per workgroup).
// How much would the K loop gain?  This program runs conv_kloop_c8's instruction stream -- same slot schedule, same LDS
// addressing (swizzled rows, zero rows, tap shifts), filter fragments streamed from L2 -- for CTW = 1 and CTW = 2 channel
// tiles per wave, back to back K loops without epilogues, 4 waves per CU on all CUs for a few seconds, and prints the
// time per unit of matrix work (one channel tile x three pixel tiles x 9 taps x 128 input channels = 324 MFMA slots).
// Round 4 added: the product's loop (csrc/xq_c8_kloop.h) with s_memtime around it, its variants without filter loads / LDS
// reads, six pixel tiles per filter fragment, and the c6 form (bf6 correction operands: FMT = 1).
//     hipcc --offload-arch=gfx950 -O3 tools/probes/c8_kloop_probe.hip -o tools/probes/c8_kloop_probe
//     tools/probes/c8_kloop_probe [seconds]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../chinesechess-alphazero_amd/csrc/xq_c8_kloop.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int C = 128, RB = 2 * C, CPR = RB / 16, SWZ = 15, NT = 3, KK = C / 16, NB = C / 64, W_RING = 4;
constexpr int ZROW = 96, ROWS = ZROW + 16, PART_BYTES = ROWS * RB, REGION = 2 * PART_BYTES;

template <int CTW>
__global__ __launch_bounds__(256, 1) void k_probe(const uint4* __restrict__ wmain, const uint4* __restrict__ wc8,
                                                 const uint4* __restrict__ image, float* __restrict__ out, int convs)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[REGION];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < REGION / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = image[i];
    __syncthreads();
    const int kb = lane >> 5, ln = lane & 31;
    constexpr int CT = 4 * CTW;                                   // channel tiles of the (pretended) layer
    const uint4* wq = wmain + (wave * CTW) * 64 + lane;
    const uint4* wc = wc8 + (size_t)(wave * CTW) * 2 * 64 + lane;
    constexpr int W_STEP = CT * 64;
    int qy[3], qx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = t * 32 + ln;
        qy[t] = q < 90 ? q / 9 : 100;
        qx[t] = q - (q / 9) * 9;
    }
    auto tap_row = [&](int dy, int dx, int p) {
        const bool ok = (unsigned)(qy[p] + dy) < 10u && (unsigned)(qx[p] + dx) < 9u;
        const int nominal = p * 32 + ln + dy * 9 + dx;
        const int row = ok ? nominal : ZROW + (nominal & 15);
        return row * RB + (((kb ^ nominal) & SWZ) << 4);
    };
    const int lane_c = (kb * 3) << 4;
    auto load_c8 = [&](int pre_p, int q, int b, int h) {
        return *reinterpret_cast<const uint4*>(lds + PART_BYTES + (pre_p ^ lane_c ^ ((q * (CPR / 2) + 4 * b + h) << 4)));
    };
    auto load_px = [&](int off) { return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(lds + off)); };
    auto load_w = [&](int step, int c) { return __builtin_bit_cast(f16x8, wq[(size_t)step * W_STEP + c * 64]); };
    auto load_wc = [&](int blk, int q, int c, int h) { return wc[(size_t)((blk * 2 + q) * CT * 2 + c * 2 + h) * 64]; };
    auto put = [&](i32x8& d, uint4 t, int h) { d[4 * h + 0] = t.x; d[4 * h + 1] = t.y; d[4 * h + 2] = t.z; d[4 * h + 3] = t.w; };

    f32x16 acc[CTW * NT];
#pragma unroll
    for (int p = 0; p < CTW * NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    for (int conv = 0; conv < convs; ++conv) {
        int pre[NT], pre_n[NT];
        f16x8 wf[W_RING][CTW];
        f16x8 px[2][NT];
        i32x8 cx[2][NT];
        i32x8 wcr[2][2][CTW];
#pragma unroll
        for (int p = 0; p < NT; ++p) pre[p] = tap_row(-1, -1, p);
#pragma unroll
        for (int s = 0; s < W_RING - 1; ++s)
#pragma unroll
            for (int c = 0; c < CTW; ++c) wf[s][c] = load_w(s, c);
#pragma unroll
        for (int p = 0; p < NT; ++p) px[0][p] = load_px(pre[p]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                put(wcr[0][q][c], load_wc(0, q, c, 0), 0);
                put(wcr[0][q][c], load_wc(0, q, c, 1), 1);
            }
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            put(cx[0][p], load_c8(pre[p], 0, 0, 0), 0);
            put(cx[0][p], load_c8(pre[p], 0, 0, 1), 1);
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int tn = tap < 8 ? tap + 1 : 8;
            const int ndy = tn / 3 - 1, ndx = tn - (tn / 3) * 3 - 1;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int blk = tap * NB + b;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int k4 = half * 2 + k2, kk = b * 4 + k4, step = tap * KK + kk;
                        const int* rows = kk + 1 < KK ? pre : pre_n;
                        const int kn = (kk + 1) % KK;
#pragma unroll
                        for (int i = 0; i < NT; ++i)
#pragma unroll
                            for (int c = 0; c < CTW; ++c) {
                                acc[c * NT + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk % W_RING][c], px[kk & 1][i],
                                                                                      acc[c * NT + i], 0, 0, 0);
                                if (c == 0) px[(kk + 1) & 1][i] = load_px(rows[i] ^ (kn << 5));
                                if (c == CTW - 1 && kk < NT) pre_n[kk] = i == NT - 1 ? tap_row(ndy, ndx, kk) : pre_n[kk];
                                if (i == 0) {
                                    wf[(kk + W_RING - 1) % W_RING][c] = load_w(step + W_RING - 1, c);
                                    put(wcr[(b + 1) & 1][k4 >> 1][c], load_wc(blk + 1, k4 >> 1, c, k4 & 1), k4 & 1);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int c = 0; c < CTW; ++c) {
                            acc[c * NT + i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                wcr[b & 1][half][c], cx[half][i], acc[c * NT + i], 0, 0, 0, 127, 0, 127);
                            if (c == 0) {
                                if (half == 0) {
                                    put(cx[1][i], load_c8(pre[i], 1, b, 0), 0);
                                    put(cx[1][i], load_c8(pre[i], 1, b, 1), 1);
                                } else {
                                    const int* r = b + 1 < NB ? pre : pre_n;
                                    const int bn = b + 1 < NB ? b + 1 : 0;
                                    put(cx[0][i], load_c8(r[i], 0, bn, 0), 0);
                                    put(cx[0][i], load_c8(r[i], 0, bn, 1), 1);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) pre[p] = pre_n[p];
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < CTW * NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[p][r];
    out[blockIdx.x * 256 + tid] = s;
}

// ---- round 4: the product's K loop (csrc/xq_c8_kloop.h), WGS workgroups of four waves per CU ----------------------------
template <int WGS, int PROBE, int FMT = 0>
__global__ __launch_bounds__(256, WGS) void k_probe_r4(const uint4* __restrict__ packed, const uint4* __restrict__ image,
                                                        float* __restrict__ out, int convs, long long* __restrict__ cycles)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[REGION];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < REGION / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = image[i];
    __syncthreads();
    const c8k::Filter flt = c8k::make_filter(packed, wave, lane);
    const c8k::Image img{0, ZROW, PART_BYTES};
    c8k::f32x16 acc[NT], sum[NT];
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[p][r] = 0.0f;
    const long long t0 = clock64();                    // s_memtime: shader cycles
    for (int conv = 0; conv < convs; ++conv) {
        c8k::kloop<NT, c8k::NoShadow, PROBE, true, 128, FMT>(lds, img, flt, lane, acc, 127 - 11, 127);
        if (conv + 1 == convs) {
#pragma unroll
            for (int p = 0; p < NT; ++p) sum[p] += acc[p];
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += sum[p][r];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 7 && tid == 0) cycles[0] = clock64() - t0;
}

// ---- what two boards per filter fragment would buy: the same loop over SIX pixel tiles (two boards in one image; one
// workgroup of four waves per CU, 512 registers: the shape k_conv3x3_c8 runs) ----
constexpr int ZROW2 = 192, PART2 = (ZROW2 + 16) * RB, REGION2 = 2 * PART2;
// (round 5: MINW = 2 = the 256-register budget of a kernel that also carries four copy waves; LOOK / RING = 1 / 1: the
//  pixel-fragment ring that fits there; FMT = 1: c6)
template <int FMT, int LOOK, int RING, int MINW>
__global__ __launch_bounds__(256 * MINW, MINW) void k_probe_r4_nt6(const uint4* __restrict__ packed, const uint4* __restrict__ image,
                                                             float* __restrict__ out, int convs, long long* __restrict__ cycles)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[REGION2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid >= 256) return;                            // (MINW = 2: four idle waves, as many registers as k_resblock_c8x2 has)
    // two copies of the one-board image (rows 0..89 and 90..179 of each part), zero rows at 192
    for (int i = tid; i < REGION2 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int part = 0; part < 2; ++part)
        for (int i = tid; i < 90 * 16; i += 256) {
            const uint4 v = image[part * (PART_BYTES / 16) + i];
            reinterpret_cast<uint4*>(lds + part * PART2)[i] = v;
            reinterpret_cast<uint4*>(lds + part * PART2)[90 * 16 + i] = v;
        }
    __syncthreads();
    const c8k::Filter flt = c8k::make_filter(packed, wave, lane);
    const c8k::Image img{0, ZROW2, PART2};
    c8k::f32x16 acc[6];
    float s = 0.0f;
    const long long t0 = clock64();
    for (int conv = 0; conv < convs; ++conv) {
        c8k::kloop<6, c8k::NoShadow, 0, true, 128, FMT, LOOK, RING>(lds, img, flt, lane, acc, 127 - 11, 127);
        if (conv + 1 == convs) {
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[p][r];
        }
    }
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 7 && tid == 0) cycles[0] = clock64() - t0;
}

static uint32_t rnd_state = 7;
static uint32_t rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

template <int CTW> static double run(const uint4* wm, const uint4* wc, const uint4* img, float* out, int blocks, int convs)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<CTW>), dim3(blocks), dim3(256), 0, 0, wm, wc, img, out, convs);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv)
{
    const double seconds = argc >= 2 ? atof(argv[1]) : 3.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    // filters: fp16 values in [-1, 1) and e4m3 bytes with the top exponent bit cleared; 8 channel tiles (CTW = 2) worth
    const size_t main_u4 = (size_t)(9 * KK + 3) * 8 * 64, c8_u4 = (size_t)(9 * NB + 1) * 2 * 8 * 2 * 64;
    std::vector<uint16_t> wm(main_u4 * 8);
    for (auto& v : wm) { _Float16 h = (_Float16)(((int)(rnd() & 0xFFFF) - 32768) / 32768.0f); memcpy(&v, &h, 2); }
    std::vector<uint8_t> wc(c8_u4 * 16);
    for (auto& v : wc) v = (uint8_t)(rnd() & 0xBF);
    std::vector<uint8_t> img(REGION, 0);
    for (int r = 0; r < 90; ++r) {
        for (int i = 0; i < C; ++i) { _Float16 h = (_Float16)((rnd() & 0xFFFF) / 65536.0f); memcpy(&img[r * RB + 2 * i], &h, 2); }
        for (int i = 0; i < RB; ++i) img[PART_BYTES + r * RB + i] = (uint8_t)(rnd() & 0x3F);
    }
    uint4 *dwm, *dwc, *dimg;
    float* out;
    CK(hipMalloc(&dwm, wm.size() * 2)); CK(hipMalloc(&dwc, wc.size())); CK(hipMalloc(&dimg, REGION));
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(dwm, wm.data(), wm.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwc, wc.data(), wc.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dimg, img.data(), REGION, hipMemcpyHostToDevice));
    {
        // the product's loop: packed filter = fp16 fragments, c8 pieces, the two scale exponents
        std::vector<uint8_t> pk((size_t)(c8k::MAIN_U4 + c8k::C8_U4 + 1) * 16 + 2 * 128, 0);     // (+ the per-row shifts)
        for (size_t i = 0; i < (size_t)c8k::MAIN_U4 * 8; ++i) memcpy(&pk[2 * i], &wm[i % wm.size()], 2);
        for (size_t i = 0; i < (size_t)c8k::C8_U4 * 16; ++i) pk[(size_t)c8k::MAIN_U4 * 16 + i] = wc[i % wc.size()];
        uint4* dpk;
        CK(hipMalloc(&dpk, pk.size()));
        CK(hipMemcpy(dpk, pk.data(), pk.size(), hipMemcpyHostToDevice));
        long long* dcyc;
        CK(hipMalloc(&dcyc, 8));
        for (int var = 0; var < 13; ++var) {
            const int wgs = var == 1 || var == 7 ? 2 : 1;      // variants: 1 WG, 2 WGs, no filter loads, no LDS reads, neither
            auto go = [&](int convs) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0));
                if (var == 0) hipLaunchKernelGGL((k_probe_r4<1, 0>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 1) hipLaunchKernelGGL((k_probe_r4<2, 0>), dim3(2 * blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 2) hipLaunchKernelGGL((k_probe_r4<1, 1>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 3) hipLaunchKernelGGL((k_probe_r4<1, 2>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 4) hipLaunchKernelGGL((k_probe_r4<1, 3>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 5) hipLaunchKernelGGL((k_probe_r4_nt6<0, 2, 3, 1>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 9) hipLaunchKernelGGL((k_probe_r4_nt6<0, 1, 1, 2>), dim3(blocks), dim3(512), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 10) hipLaunchKernelGGL((k_probe_r4_nt6<1, 2, 3, 1>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 11) hipLaunchKernelGGL((k_probe_r4_nt6<1, 1, 1, 2>), dim3(blocks), dim3(512), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 12) hipLaunchKernelGGL((k_probe_r4_nt6<1, 1, 1, 1>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 6) hipLaunchKernelGGL((k_probe_r4<1, 0, 1>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else if (var == 7) hipLaunchKernelGGL((k_probe_r4<2, 0, 1>), dim3(2 * blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                else hipLaunchKernelGGL((k_probe_r4<1, 3, 1>), dim3(blocks), dim3(256), 0, 0, dpk, dimg, out, convs, dcyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0.0f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                return (double)ms;
            };
            go(20);
            const double probe = go(200);
            const int chunk = (int)(200 * 500.0 / probe);
            double last = 0.0;
            for (int i = 0; i < (int)(seconds / 0.5) + 1; ++i) last = go(chunk);
            long long cyc = 0;
            CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
            const char* what[13] = {"as in the product", "2 workgroups per CU", "no filter loads in the loop", "no LDS reads in the loop",
                                   "no loads at all in the loop",
                                   "six pixel tiles = TWO boards per filter fragment (halve the figures for one board)",
                                   "c6: bf6 correction operands, MFMA floor 10368", "c6, 2 workgroups per CU", "c6, no loads at all in the loop",
                                   "c8 six tiles, ring 1/1, 256-register budget (halve for one board)",
                                   "c6 six tiles, ring 2/3, 512 registers (halve for one board)",
                                   "c6 six tiles, ring 1/1, 256-register budget (halve for one board)",
                                   "c6 six tiles, ring 1/1, 512 registers (halve for one board)"};
            printf("RESULT r4 loop (%s): %.2f us per K loop of a wave, %.2f us per K loop and CU; %.0f shader cycles per K loop "
                   "(MFMA floor 13824) -> %.2f GHz\n", what[var], last * 1e3 / chunk, last * 1e3 / chunk / wgs,
                   (double)cyc / chunk, (double)cyc / chunk / (last * 1e3 / chunk) / 1e3);
        }
    }
    for (int ctw = 1; ctw <= 2; ++ctw) {
        auto go = [&](int convs) { return ctw == 1 ? run<1>(dwm, dwc, dimg, out, blocks, convs) : run<2>(dwm, dwc, dimg, out, blocks, convs); };
        go(20);
        const double probe = go(200);
        const int chunk = (int)(200 * 500.0 / probe);
        double last = 0.0;
        for (int i = 0; i < (int)(seconds / 0.5) + 1; ++i) last = go(chunk);
        const double us_per_conv = last * 1e3 / chunk;
        printf("RESULT ctw %d: %.2f us per K loop of a wave (%d channel tile(s) x 3 pixel tiles x 9 taps), %.2f us per channel tile"
               " -- the product's k_resblock<C8> spends ~%.1f us per convolution and board at 3.3 ms per block\n",
               ctw, us_per_conv, ctw, us_per_conv / ctw, 3.3e3 / 2.0 / 128.0);
    }
    return 0;
}
