// xq_conv.hip -- the convolutions of the policy/value ResNet as hand-written MFMA kernels (gfx950).
//
//   conv_kloop      the shared K loop (9 taps x C channels out of an LDS image, weights streamed from L2)
//   k_conv3x3       kernel 1: one convolution per launch, P boards per workgroup (any supported filter count / mode)
//   k_resblock      kernel 2: a whole residual block per launch, persistent, matrix waves + copy waves; optionally the
//                   two 1x1 head convolutions folded into the last block's store pass
//   k_input_conv    kernel 3: the 5x5 input convolution on the feature planes as the search kernel writes them
//   k_split_bias_act  fp32 -> operand pair (used when the input layer comes from a library convolution)
//
// Reference: the residual tower of CChessModel.build / _build_residual_block (cchess_alphazero/agent/model.py:40-83):
// Conv2D(F, 3, padding="same", use_bias=False) -> BatchNorm -> (+ skip) -> ReLU on 10x9 boards.  With BatchNorm folded
// into the weights this is  y = act(conv3x3(x, w) + bias (+ skip))  and it is 97 % of the FLOPs of a self-play round.
//
// Design (not an im2col GEMM, not a library call):
//   * one workgroup = P whole boards.  Their activations ([90 pixels][C] channels-last, 2-byte elements) are copied
//     ONCE into LDS; all 9 taps x C input channels are then read from that LDS image, so HBM/L2 sees every
//     activation exactly once per layer.  Pixel rows are XOR-swizzled by (row & 15) in 16-byte chunks so that the
//     32 rows a ds_read_b128 touches fall on 16 distinct bank slots per lane group; out-of-board taps point at an
//     all-zero row instead of being predicated.
//   * wave w owns output channels [32w, 32w+32) for every pixel of the P boards: v_mfma_f32_32x32x16 with the
//     WEIGHTS as the A operand (rows = output channels) and the PIXELS as the B operand (columns = pixels), so each
//     lane ends up with 4 runs of 4 consecutive channels of one pixel -> 8-byte channels-last stores.
//     The weights are pre-packed on the host in exactly fragment order (cz_conv3x3_pack_weights): every weight load
//     is one fully coalesced 1 KiB global_load_dwordx4 per wave, served from L2 (the whole packed filter is < 1.2 MiB),
//     prefetched three K-steps ahead through a 4-deep register ring.  The K loop has no barrier at all.
//   * precision.  parts = 1: operands are bf16 (or fp16), fp32 accumulate.  parts = 2 ("split" mode): every operand is
//     a (hi, lo) pair of bf16 with hi + lo equal to the fp32 value to 2^-17, and the kernel accumulates
//     hi*hi + hi*lo + lo*hi in fp32 -- three bf16 MFMAs (1/16 the cost of an fp32 MFMA each) give a product error of
//     ~1e-5 relative, which keeps policy/value within the 1e-4 the parity contract asks for while running at several
//     times the fp32 matrix rate.  The epilogue re-splits its fp32 result into (hi, lo) for the next layer.
//   * epilogue fused: + bias, + skip (read as hi + lo), ReLU, split / convert, store.  The last trunk layer can
//     emit fp32 directly (y_f32) for the policy/value heads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include "../../include/czero.h"
#include "xq_c8_kloop.h"
#include "xq_nn_common.h"

extern "C" void czi_set_error(const char* msg);
// csrc/xq_tower.hip: a chain of (hi, lo) pair blocks on four matrix waves (k_tower_pairs4<E, 128 / 192>)
extern "C" int czi_pairs4_launch(const void* x_hi, const void* x_lo, int n_blocks, const void* const* w1, const float* const* b1,
                                 const void* const* w2, const float* const* b2, void* y_hi, void* y_lo, const float* head_w,
                                 const float* head_b, float* pol, float* val, int n_pol, int n_boards, int channels, int dtype,
                                 int n_cu, const int32_t* n_dev, void* stream, float* y_f32);

namespace {

// Compact evaluation queue (cz_*_q entry points): the number of boards comes from DEVICE memory (no host
// synchronisation, the launch shape stays fixed so that a round can be replayed from a HIP graph) and the input
// convolution gathers its planes through a row list.  Set by the _q entry points around the ordinary dispatch code.
struct QueueCtx {
    const int32_t* rows = nullptr;      // [n] queue slot of compact board i (input convolution only)
    const int32_t* n_dev = nullptr;     // [1] boards to process (<= the n_boards argument)
};
thread_local QueueCtx g_q;

// ---- geometry shared by both kernels -----------------------------------------------------------------------------
template <int C, int P, int PARTS> struct Geom {
    static constexpr int RB = C * 2;                 // bytes per pixel row in LDS
    static constexpr int CPR = RB / 16;              // 16-byte chunks per row
    static constexpr bool POW2 = (RB & (RB - 1)) == 0;        // 192 filters: 384-byte rows, 24 chunks
    // swizzle: chunk ^= row & SWZ.  It must stay inside the row: all chunks for power-of-two rows, an aligned group of 8
    // otherwise (two of the 16 rows of a read group then share a slot: a 2-way conflict instead of none)
    static constexpr int SWZ = POW2 ? (CPR - 1 < 15 ? CPR - 1 : 15) : 7;
    static_assert(POW2 || CPR % 8 == 0, "rows must be a whole number of 8-chunk groups");
    // byte offset of K-step kk relative to pre[] (= row * RB + ((kb ^ (row & SWZ)) << 4)): the chunk is 2 * kk + kb and
    // the XOR only touches the bits SWZ covers, the rest of 2 * kk is a plain add
    static __device__ __forceinline__ int kstep(int pre, int kk)
    {
        if (POW2) return pre ^ (kk << 5);
        return (pre ^ ((kk & 3) << 5)) + ((kk >> 2) << 7);
    }
    // 16 all-zero rows for out-of-board taps and padding pixels.  A lane whose tap leaves the board reads zero row
    // ZROW + (r & 15), r = the row it would have read: every ds_read_b128 lane group (16 lanes, MI355X_MICROARCH.md
    // section LDS) covers 16 consecutive values of r, so its 16 lanes -- on the board or not -- keep 16 distinct
    // (row & 15), i.e. 16 distinct 16-byte slots of the 256-byte bank row.  (One shared zero row cost a 2-way conflict in
    // almost every group that had an off-board lane: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.36 in round 1.)
    static constexpr int ZROW = (P * 90 + 15) / 16 * 16;
    static constexpr int ZROWS = 16;
    static constexpr int PART_BYTES = (ZROW + ZROWS) * RB;
    static constexpr int REGION = PARTS * PART_BYTES;
    static constexpr int NT = P * 3;                 // pixel tiles of 32 (96 slots per board, 90 used)
    static constexpr int KK = C / 16;                // K-steps per tap
    static constexpr int CT = C / 32;                // channel tiles = waves per board group
    static constexpr int GTHREADS = CT * 64;
    static constexpr int CHUNKS = P * 90 * CPR;      // 16-byte chunks per part
    static constexpr int ITER = (CHUNKS + GTHREADS - 1) / GTHREADS;
    static constexpr int W_STEP = CT * 64;           // uint4 per K-step of packed weights
    static constexpr int W_RING = KK < W_RING_MAX ? KK : W_RING_MAX;   // ring slot = step % W_RING
    static constexpr int W_PART = (9 * KK + W_PAD_STEPS) * W_STEP;
    static_assert(KK % W_RING == 0 && KK % 2 == 0, "ring / double-buffer indices are taken from kk");
};

// global -> registers: the P boards starting at board n0 (16 bytes per lane per iteration, fully coalesced)
template <typename E, int C, int P, int PARTS, int NTHR = Geom<C, P, PARTS>::GTHREADS>
__device__ __forceinline__ void tile_load(const E* xh, const E* xl, int n0, int n_boards, int gtid,
                                          uint4 (*v)[(Geom<C, P, PARTS>::CHUNKS + NTHR - 1) / NTHR])
{
    typedef Geom<C, P, PARTS> G;
    constexpr int ITER = (G::CHUNKS + NTHR - 1) / NTHR;
    const bool full = n0 + P <= n_boards;
#pragma unroll
    for (int part = 0; part < PARTS; ++part) {
        const uint4* src = reinterpret_cast<const uint4*>((part ? xl : xh) + (size_t)n0 * 90 * C);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = it * NTHR + gtid;
            const bool ok = ((it + 1) * NTHR <= G::CHUNKS || i < G::CHUNKS) &&
                            (full || n0 + i / (G::CPR * 90) < n_boards);
            v[part][it] = make_uint4(0, 0, 0, 0);
            if (ok) v[part][it] = src[i];
        }
    }
}

// registers -> LDS image: [row][chunk ^ (row & SWZ)], plus the all-zero row
template <int C, int P, int PARTS, int NTHR = Geom<C, P, PARTS>::GTHREADS>
__device__ __forceinline__ void tile_write(unsigned char* region, int gtid,
                                           const uint4 (*v)[(Geom<C, P, PARTS>::CHUNKS + NTHR - 1) / NTHR])
{
    typedef Geom<C, P, PARTS> G;
    constexpr int ITER = (G::CHUNKS + NTHR - 1) / NTHR;
#pragma unroll
    for (int part = 0; part < PARTS; ++part) {
        unsigned char* dst = region + part * G::PART_BYTES;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = it * NTHR + gtid;
            const int row = i / G::CPR, ch = i % G::CPR;
            if ((it + 1) * NTHR <= G::CHUNKS || i < G::CHUNKS)
                *reinterpret_cast<uint4*>(dst + row * G::RB + ((ch ^ (row & G::SWZ)) << 4)) = v[part][it];
        }
    }
}

// the 16 all-zero rows of an image (written once per workgroup: nothing else ever stores there)
template <int C, int P, int PARTS, int NTHR = Geom<C, P, PARTS>::GTHREADS>
__device__ __forceinline__ void zero_rows_write(unsigned char* region, int gtid)
{
    typedef Geom<C, P, PARTS> G;
#pragma unroll
    for (int part = 0; part < PARTS; ++part)
        for (int i = gtid; i < G::ZROWS * G::CPR; i += NTHR)
            *reinterpret_cast<uint4*>(region + part * G::PART_BYTES + G::ZROW * G::RB + i * 16) = make_uint4(0, 0, 0, 0);
}

// The K loop: 9 taps x C input channels for the 32 output channels of this wave and all NT pixel tiles of the
// LDS image.  acc[p][r] <-> pixel (p % 3) * 32 + (lane & 31) of board p / 3,
//                          channel 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
//   row_base / zrow / part_bytes: where the image's first row, the 16 zero rows and the second operand part are, for
//   kernels whose images share one LDS allocation (k_resblock_ip); the defaults are the self-contained layout of Geom.
//   CTW: channel tiles per wave (wq points at the wave's FIRST tile; the tiles of a K-step are 64 uint4 apart).  Every
//   pixel fragment read from LDS then feeds CTW MFMAs: with plain operands and one tile per wave there is one
//   ds_read_b128 (8 LDS cycles) per MFMA (32 cycles of one of four SIMDs), i.e. the LDS port is as busy as the matrix
//   pipes; two tiles per wave halve that.  acc[c * NT + p] <-> channel tile c of the wave.
template <typename E, int C, int P, int PARTS, int CTW = 1>
__device__ __forceinline__ void conv_kloop(const unsigned char* region, const uint4* wq, int lane,
                                           f32x16* acc, int row_base = 0, int zrow = Geom<C, P, PARTS>::ZROW,
                                           int part_bytes = Geom<C, P, PARTS>::PART_BYTES)
{
    typedef Geom<C, P, PARTS> G;
    typedef typename Mfma<E>::V8 V8;
    constexpr int NT = G::NT, KK = G::KK, W_RING = G::W_RING;
    static_assert(CTW == 1 || PARTS == 1, "several channel tiles per wave: plain operands only");
    const int kb = lane >> 5, ln = lane & 31;
    // byte offset (within a part) of this lane's 16-byte fragment piece for K-step 0 of a tap: row*RB + swizzle bits.
    // chunk = 2*kk + kb, swizzled chunk = chunk ^ (row & SWZ) = (2*kk) ^ (kb ^ (row & SWZ)): the lane part is folded
    // into pre[], the K-step part is one XOR with the constant kk << 5.
    int pre[NT], pre_n[NT];
    int qy[3], qx[3];                                  // board row / column of this lane's pixel in each of the 3 tiles
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = t * 32 + ln;
        qy[t] = q < 90 ? q / 9 : 100;                  // 100: never on the board, whatever the tap
        qx[t] = q - (q / 9) * 9;
    }
    auto tap_row = [&](int dy, int dx, int p) {
        const int t = p % 3;
        const bool ok = (unsigned)(qy[t] + dy) < 10u && (unsigned)(qx[t] + dx) < 9u;
        const int nominal = (p / 3) * 90 + t * 32 + ln + dy * 9 + dx;
        const int row = ok ? row_base + nominal : zrow + (nominal & 15);
        // (swizzle key = the image-relative row; zrow is a multiple of 16, so a zero row keys like the row it stands for)
        return row * G::RB + (((kb ^ nominal) & G::SWZ) << 4);
    };
    constexpr int WP = PARTS * CTW;                    // weight fragments per K-step: [part] (CTW == 1) or [tile]
    V8 wf[W_RING][WP];
    V8 px[2][NT][PARTS];
#pragma unroll
    for (int p = 0; p < CTW * NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    auto load_w = [&](int step, int f) {               // f: part (CTW == 1) or channel tile of the wave
        const int part = CTW == 1 ? f : 0, c = CTW == 1 ? 0 : f;
        return __builtin_bit_cast(V8, wq[(size_t)part * G::W_PART + (size_t)step * G::W_STEP + c * 64]);
    };
    auto load_px = [&](int off, int part) {
        return __builtin_bit_cast(V8, *reinterpret_cast<const uint4*>(region + part * part_bytes + off));
    };

#pragma unroll
    for (int p = 0; p < NT; ++p) pre[p] = tap_row(-1, -1, p);
#pragma unroll
    for (int s = 0; s < W_RING - 1; ++s)
#pragma unroll
        for (int f = 0; f < WP; ++f) wf[s][f] = load_w(s, f);
#pragma unroll
    for (int part = 0; part < PARTS; ++part)
#pragma unroll
        for (int p = 0; p < NT; ++p) px[0][p][part] = load_px(pre[p], part);

    // One K-step = NT (x3 in split mode) MFMAs.  The LDS reads of the NEXT K-step and the weight loads three K-steps
    // ahead are issued one per MFMA, in the shadow of the matrix pipe; sched_barrier pins that order (left alone, the
    // compiler sinks every load to just before its use and the pipe drains at each K-step).
    constexpr int NM = CTW * NT * (PARTS == 2 ? 3 : 1);
    constexpr int NL = NT * PARTS;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        // rows of the NEXT tap: computed a few at a time in the shadow of this tap's MFMAs (done in one block at the
        // tap boundary they leave the matrix pipe idle for ~400 cycles per tap)
        const int tn = tap < 8 ? tap + 1 : 8;
        const int ndy = tn / 3 - 1, ndx = tn - (tn / 3) * 3 - 1;
        constexpr int PER = (NT + KK - 2) / (KK - 1);    // rows to prepare per K-step: all done before the last step
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int step = tap * KK + kk;
            const V8* w = wf[kk % W_RING];
            V8 (*b)[PARTS] = px[kk & 1];
            V8 (*bn)[PARTS] = px[(kk + 1) & 1];
            const int* rows = kk + 1 < KK ? pre : pre_n;
            const int kn = (kk + 1) % KK;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int pass = i / NT, p = i % NT;      // pass 0: w_hi*x_hi, 1: w_lo*x_hi, 2: w_hi*x_lo
                if (CTW == 1) acc[p] = Mfma<E>::mma(w[pass == 1 ? PARTS - 1 : 0], b[p][pass == 2 ? PARTS - 1 : 0], acc[p]);
                else acc[i] = Mfma<E>::mma(w[pass], b[p][0], acc[i]);      // pass = channel tile of the wave
                if (i < NL) bn[i % NT][i / NT] = load_px(G::kstep(rows[i % NT], kn), i / NT);
                if (i >= NM - PER && kk * PER + (i - (NM - PER)) < NT)
                    pre_n[kk * PER + (i - (NM - PER))] = tap_row(ndy, ndx, kk * PER + (i - (NM - PER)));
                if (i >= NM - WP)
                    wf[(kk + W_RING - 1) % W_RING][i - (NM - WP)] = load_w(step + W_RING - 1, i - (NM - WP));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int p = 0; p < NT; ++p) pre[p] = pre_n[p];
    }
}

// ---- kernel 1: one workgroup = P boards, one pass (any channel count; used for the small / plain-precision cases) --
template <typename E, int C, int P, int PARTS, int MINW>
__global__ __launch_bounds__(C / 32 * 64, MINW) void k_conv3x3(
    const E* __restrict__ xh, const E* __restrict__ xl, const E* __restrict__ wp, const float* __restrict__ bias,
    const E* __restrict__ sh, const E* __restrict__ sl, E* __restrict__ yh, E* __restrict__ yl,
    float* __restrict__ yf, int n_boards, int relu)
{
    typedef Geom<C, P, PARTS> G;
    constexpr int NT = G::NT;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::REGION];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * P;
    {
        uint4 v[PARTS][G::ITER];
        tile_load<E, C, P, PARTS>(xh, xl, n0, n_boards, tid, v);
        tile_write<C, P, PARTS>(lds, tid, v);
        zero_rows_write<C, P, PARTS>(lds, tid);
    }
    __syncthreads();

    f32x16 acc[NT];
    conv_kloop<E, C, P, PARTS>(lds, reinterpret_cast<const uint4*>(wp) + wave * 64 + lane, lane, acc);

    // ---- epilogue: lane holds, for pixel (tile p, column ln), channels 32*wave + 8*g + 4*kb + {0..3}, g = 0..3 ----
    const int kb = lane >> 5, ln = lane & 31;
    // everything the epilogue reads from memory is requested up front (round 6: with a load next to every store the compiler
    // waited for each -- 4 NT bias + skip round trips in a row; a one-board batch, the `mini` configuration, felt every one)
    f32x4 bq[4];
    c8k::u32x2 skh[NT][4], skl[NT][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(bias + wave * 32 + g * 8 + kb * 4);
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        const int q = (p % 3) * 32 + ln;
        const int n = n0 + p / 3;
        const bool live = q < 90 && n < n_boards;
        const size_t pix = ((size_t)(live ? n : 0) * 90 + (live ? q : 0)) * C;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wave * 32 + g * 8 + kb * 4;
            skh[p][g] = c8k::u32x2{0u, 0u};
            skl[p][g] = c8k::u32x2{0u, 0u};
            if (sh && live) {
                skh[p][g] = *reinterpret_cast<const c8k::u32x2*>(sh + pix + ch);
                if (PARTS == 2) skl[p][g] = *reinterpret_cast<const c8k::u32x2*>(sl + pix + ch);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[g]));
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        const int q = (p % 3) * 32 + ln;
        const int n = n0 + p / 3;
        if (q >= 90 || n >= n_boards) continue;
        const size_t pix = ((size_t)n * 90 + q) * C;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wave * 32 + g * 8 + kb * 4;
            const f32x4 bv = bq[g];
            float v[4] = {acc[p][g * 4 + 0] + bv[0], acc[p][g * 4 + 1] + bv[1], acc[p][g * 4 + 2] + bv[2],
                          acc[p][g * 4 + 3] + bv[3]};
            if (sh) {
                const Quad<E> a = __builtin_bit_cast(Quad<E>, skh[p][g]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += (float)a.e[i];
                if (PARTS == 2) {
                    const Quad<E> b2 = __builtin_bit_cast(Quad<E>, skl[p][g]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)b2.e[i];
                }
            }
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
            }
            if (yf) {
                *reinterpret_cast<float4*>(yf + pix + ch) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                Quad<E> hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi.e[i] = (E)v[i];
                    lo.e[i] = (E)(v[i] - (float)hi.e[i]);
                }
                *reinterpret_cast<Quad<E>*>(yh + pix + ch) = hi;
                if (PARTS == 2) *reinterpret_cast<Quad<E>*>(yl + pix + ch) = lo;
            }
        }
    }
}

// ---- the c8 tower arithmetic: fp16 main term + two block-scaled fp8 correction terms --------------------------------
// The tower is bound by the socket's power cap and spends three bf16 MFMAs per product.  With an fp16 main
// term (11 bits) the correction terms  w_hi x_lo  and  w_lo x_hi  need four significant bits, which is what an e4m3 operand
// holds:   w x  ~  f16(w) f16(x)  +  e4m3(w) e4m3(x - f16(x))  +  e4m3(w - f16(w)) e4m3(x)
// as v_mfma_f32_32x32x16_f16 + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 per 64 input channels: 2.0 instead of 3.0 MFMA-equivalents
// per product (measured 1.47x in register-resident loops, profiles/r03_fp8_corrections_study.json; per-product accuracy
// 2^-16, policy within 2e-5 of float64 in the emulated 7 x 128 network, same file).  k_conv3x3_c8 is the single-convolution
// form (cz_conv3x3_c8); k_resblock<..., C8> (dtype CZ_F16C8) is the residual block the self-play engine runs by default,
// k_input_conv<..., C8> produces its operands; the pipelined block (k_resblock_pipe) still computes the bf16 arithmetic.
//   operands: x_hi f16 [n][90][C];  x_c8 bytes [n][90][2C] = e4m3(x_lo * 2^11) for the C channels, then e4m3(x) for them
//   LDS:      part 0 = x_hi rows, part 1 = x_c8 rows (same 2C bytes per pixel: the image code of the split kernels serves)
//   weights:  f16 fragments as in cz_conv3x3_pack_weights, then per (tap, 64-channel block, kind) two 16-byte pieces per
//             lane: kind 0 = e4m3(w * 2^sh) (meets x_lo), kind 1 = e4m3((w - f16(w)) * 2^sl) (meets e4m3(x)); lane l holds
//             output channel l % 32, input channels 32 (l / 32) .. + 31 of the block -- the same k map as the pixels;
//             the power-of-two scales sh, sl follow the fragments
template <int C, int P>
__global__ __launch_bounds__(C / 32 * 64, 1) void k_conv3x3_c8(
    const _Float16* __restrict__ xh, const unsigned char* __restrict__ xc, const uint4* __restrict__ wp,
    const float* __restrict__ bias, const _Float16* __restrict__ sh, const unsigned char* __restrict__ sc8,
    _Float16* __restrict__ yh, unsigned char* __restrict__ yc, float* __restrict__ yf, int n_boards, int relu)
{
    typedef Geom<C, P, 2> G;
    constexpr int NT = G::NT;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::REGION];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * P;
    {
        uint4 v[2][G::ITER];
        tile_load<_Float16, C, P, 2>(xh, reinterpret_cast<const _Float16*>(xc), n0, n_boards, tid, v);
        tile_write<C, P, 2>(lds, tid, v);
        zero_rows_write<C, P, 2>(lds, tid);
    }
    __syncthreads();
    static_assert(cf8::Pack<C>::MAIN_U4 == c8k::Geo<C>::MAIN_U4 && cf8::Pack<C>::C8_U4 == c8k::Geo<C>::C8_U4, "packed layout");
    // The accumulators start at bias (+ the skip operand, hi + lo8 * 2^-11): the products are added on top and the epilogue
    // has only the ReLU and the split left.  Same order in k_resblock_c8, so the two stay bit-identical.
    const int kb = lane >> 5, ln = lane & 31;
    f32x16 acc[NT];
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        const int q = (p % 3) * 32 + ln;
        const int n = n0 + p / 3;
        const bool live = q < 90 && n < n_boards;
        const size_t pixel = (size_t)(live ? n : 0) * 90 + (live ? q : 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wave * 32 + g * 8 + kb * 4;
            const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
            float v[4] = {bv.x, bv.y, bv.z, bv.w};
            if (sh && live)
                cf8::add_pair4(v, *reinterpret_cast<const Quad<_Float16>*>(sh + pixel * C + ch),
                              *reinterpret_cast<const uint32_t*>(sc8 + pixel * 2 * C + ch));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[p][g * 4 + i] = v[i];
        }
    }
    c8k::kloop<NT, c8k::NoShadow, 0, false, C>(lds, c8k::Image{0, G::ZROW, G::PART_BYTES}, c8k::make_filter<C>(wp, wave, lane),
                                               lane, acc, 127 - cf8::X_LO_SHIFT, 127);
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        const int q = (p % 3) * 32 + ln;
        const int n = n0 + p / 3;
        if (q >= 90 || n >= n_boards) continue;
        const size_t pixel = (size_t)n * 90 + q;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = wave * 32 + g * 8 + kb * 4;
            float v[4] = {acc[p][g * 4 + 0], acc[p][g * 4 + 1], acc[p][g * 4 + 2], acc[p][g * 4 + 3]};
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
            }
            if (yf) {
                *reinterpret_cast<float4*>(yf + pixel * C + ch) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                const cf8::Split4 o = cf8::split4(v);
                *reinterpret_cast<Quad<_Float16>*>(yh + pixel * C + ch) = o.hi;
                *reinterpret_cast<uint32_t*>(yc + pixel * 2 * C + ch) = o.l8;
                *reinterpret_cast<uint32_t*>(yc + pixel * 2 * C + C + ch) = o.h8;
            }
        }
    }
}

// ---- kernel 2: a whole residual block per launch, wave-specialised ---------------------------------------------------
// y = relu(conv2(relu(conv1(x) + b1)) + b2 + x) for P boards at a time per workgroup, persistent over boards.
// The intermediate activation never leaves LDS (it is written straight into a second operand image), the skip
// operand is read back from the first image, and HBM sees one read of x and one write of y per block instead of
// 2 reads + 1 skip read + 2 writes.  Waves 0..CT-1 ("matrix waves", 32 output channels each) only run K loops and the
// two register-level epilogues; 4 more waves ("copy waves") own all global traffic: they fetch the NEXT boards into
// registers while the matrix waves work (their vmcnt is their own, so nothing in the K loop waits for it), drop them
// into the X image the moment the matrix waves are done with it, and stream the PREVIOUS boards' staged result out
// (16 bytes per lane, re-split into hi/lo in split mode) under the next K loop.
//   LDS: X image | Y image | staging (fp32 in split mode, final 2-byte values otherwise); one workgroup per CU:
//     128 filters split  (P = 1): 46.6 + 46.6 + 46.1 KB, 4 + 4 waves      128 filters plain (P = 2): the same bytes
//     256 filters plain  (P = 1): 46.6 + 46.6 + 46.1 KB, 8 + 4 waves      192 filters plain (P = 1): 34.9 + 34.9 + 34.6
//     (192 / 256 filters with split operands do not fit: those run one k_conv3x3 per convolution)
//   barriers per tile: A (X ready) .. K1 .. epi1 -> Y .. B (Y ready) .. K2 .. epi2 -> staging .. C (staged, X free)
constexpr int RB_COPY_THREADS = 256;

template <typename E, int C, int PARTS, int P, bool HEADS = false, int CTW = 1>
__global__ __launch_bounds__((C / 32 / CTW + 4) * 64, (C / 32 / CTW + 4 + 3) / 4) void k_resblock(
    const E* __restrict__ xh, const E* __restrict__ xl, const E* __restrict__ w1p, const float* __restrict__ b1,
    const E* __restrict__ w2p, const float* __restrict__ b2, E* __restrict__ yh, E* __restrict__ yl,
    float* __restrict__ yf, int n_boards, HeadArgs hd, const int32_t* __restrict__ n_dev)
{
    static_assert(!HEADS || (PARTS == 2 && C / 8 == 16), "fused heads: split operands, 128 filters");
    if (n_dev) {                                        // compact queue: the board count lives on the device
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    typedef Geom<C, P, PARTS> G;
    constexpr int NT = G::NT, CT = G::CT / CTW, CTHR = RB_COPY_THREADS;      // CT: matrix waves, CTW channel tiles each
    static_assert(G::CT % CTW == 0, "channel tiles per wave must divide the channel tiles");
    constexpr int SROW = PARTS == 2 ? C * 4 : C * 2;   // staging row (one pixel): fp32, or the final 2-byte values
    constexpr int PIECES = P * 90 * (C / 8);           // 8-channel output pieces per tile
    constexpr int EITER = (PIECES + CTHR - 1) / CTHR;
    constexpr int LITER = (G::CHUNKS + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G::REGION + P * 90 * SROW];
    unsigned char* X = lds;
    unsigned char* Y = lds + G::REGION;
    unsigned char* S = lds + 2 * G::REGION;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool copy_role = wave >= CT;
    const int n_tiles = (n_boards + P - 1) / P;
    int t = blockIdx.x;                                 // tile = P consecutive boards
    if (t >= n_tiles) return;
    const int stride = gridDim.x;
    // staging offset of channels ch..ch+3 of tile pixel qq
    auto stage_off = [&](int qq, int ch) {
        if (PARTS == 2) return qq * SROW + ((((ch >> 2) & ~7) | (((ch >> 2) ^ qq) & 7)) << 4);
        return qq * SROW + ((((ch >> 3) ^ (qq & G::SWZ))) << 4) + (ch & 7) * 2;
    };

    if (copy_role) {
        const int ctid = tid - CT * 64;
        uint4 v[PARTS][LITER];
        tile_load<E, C, P, PARTS, CTHR>(xh, xl, t * P, n_boards, ctid, v);
        tile_write<C, P, PARTS, CTHR>(X, ctid, v);
        zero_rows_write<C, P, PARTS, CTHR>(X, ctid);
        zero_rows_write<C, P, PARTS, CTHR>(Y, ctid);
        int t_prev = -1;
        float hw[HEADS ? 6 : 1][8];                            // this thread's slice of the head filters (c8 is fixed)
        if (HEADS) {
#pragma unroll
            for (int o = 0; o < 6; ++o)
#pragma unroll
                for (int k = 0; k < 8; ++k) hw[o][k] = hd.w[o * C + (ctid % (C / 8)) * 8 + k];
        }
        for (;;) {
            __syncthreads();                                   // A: X holds tile t
            const int tn = t + stride;
            const bool has_next = tn < n_tiles;
            int ct2 = ctid;
            asm volatile("" : "+v"(ct2));                      // keep address arithmetic inside the loop (registers)
            if (has_next) tile_load<E, C, P, PARTS, CTHR>(xh, xl, tn * P, n_boards, ct2, v);
            // stream out the previous tile (staged, already ReLU'd) while the matrix waves run K1
            auto store_tile = [&](int to) {
                const size_t ebase = (size_t)to * P * 90 * C;
                const int valid = (n_boards - to * P < P ? n_boards - to * P : P) * 90 * (C / 8);
#pragma unroll
                for (int it = 0; it < EITER; ++it) {
                    const int i = it * CTHR + ct2;
                    if (!((it + 1) * CTHR <= PIECES || i < PIECES) || i >= valid) continue;
                    const int qq = i / (C / 8), c8 = i % (C / 8);
                    if (PARTS == 2) {
                        const float4 f0 = *reinterpret_cast<const float4*>(S + stage_off(qq, c8 * 8));
                        const float4 f1 = *reinterpret_cast<const float4*>(S + stage_off(qq, c8 * 8 + 4));
                        const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                        if (HEADS) {
                            // 16 consecutive lanes hold the 128 channels of one pixel: partial dot products, then a
                            // 16-lane butterfly; lane o of the group writes head output o
                            float hs[6];
#pragma unroll
                            for (int o = 0; o < 6; ++o) {
                                float a = 0.0f;
#pragma unroll
                                for (int k = 0; k < 8; ++k) a += r[k] * hw[o][k];
                                a = row16_sum(a);
                                hs[o] = a;
                            }
                            const int board = to * P + qq / 90, q = qq % 90;
#pragma unroll
                            for (int o = 0; o < 6; ++o)
                                if (c8 == o) {
                                    float v = hs[o] + hd.b[o];
                                    v = v > 0.0f ? v : 0.0f;
                                    if (o < hd.n_pol) hd.pol[(size_t)board * (hd.n_pol * 90) + o * 90 + q] = v;
                                    else hd.val[(size_t)board * ((6 - hd.n_pol) * 90) + (o - hd.n_pol) * 90 + q] = v;
                                }
                        } else if (yf) {
                            float4* o = reinterpret_cast<float4*>(yf + ebase) + 2 * i;
                            o[0] = f0;
                            o[1] = f1;
                        } else {
                            struct alignas(16) E8 { E e[8]; };
                            E8 hi, lo;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                hi.e[k] = (E)r[k];
                                lo.e[k] = (E)(r[k] - (float)hi.e[k]);
                            }
                            reinterpret_cast<uint4*>(yh + ebase)[i] = __builtin_bit_cast(uint4, hi);
                            reinterpret_cast<uint4*>(yl + ebase)[i] = __builtin_bit_cast(uint4, lo);
                        }
                    } else {
                        const uint4 f = *reinterpret_cast<const uint4*>(S + stage_off(qq, c8 * 8));
                        reinterpret_cast<uint4*>(yh + ebase)[i] = f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (t_prev >= 0) store_tile(t_prev);
            __syncthreads();                                   // B
            __syncthreads();                                   // C: tile t is staged, X is free
            if (has_next) tile_write<C, P, PARTS, CTHR>(X, ct2, v);
            t_prev = t;
            if (!has_next) {
                store_tile(t);
                break;
            }
            t = tn;
        }
        return;
    }

    // ---- matrix waves ----
    const int wg = wave * CTW;                                 // first channel tile of this wave
    const uint4* wq1 = reinterpret_cast<const uint4*>(w1p) + wg * 64 + lane;
    const uint4* wq2 = reinterpret_cast<const uint4*>(w2p) + wg * 64 + lane;
    const int kb = lane >> 5, ln = lane & 31;
    for (;;) {
        __syncthreads();                                       // A
        const bool has_next = t + stride < n_tiles;
        f32x16 acc[CTW * NT];
        __builtin_amdgcn_s_setprio(3);
        conv_kloop<E, C, P, PARTS, CTW>(X, wq1, lane, acc);
        __builtin_amdgcn_s_setprio(0);
        // the biases of this wave's channels: all loads in flight at once, materialised once (round 6: left to the compiler there
        // was one load and one vmcnt(0) per 8-byte store -- 4 CTW NT L2 round trips in a row per epilogue, 7 % of the deep tower)
        f32x4 bq[CTW][4];
        auto bias_fetch = [&](const float* bp) {
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[c][g] = *reinterpret_cast<const f32x4*>(bp + (wg + c) * 32 + g * 8 + kb * 4);
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[c][g]));
        };
        bias_fetch(b1);
        int ln2 = ln, kb2 = kb, gt2 = tid;
        asm volatile("" : "+v"(ln2), "+v"(kb2), "+v"(gt2));
        // epilogue 1: relu(acc + b1) -> (hi, lo) -> Y image (operand layout of the second convolution)
#pragma unroll
        for (int cp = 0; cp < CTW * NT; ++cp) {
            const int p = cp % NT, c = cp / NT;
            const int q = (p % 3) * 32 + ln2;
            const int row = (p / 3) * 90 + q;
            if (q < 90) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = (wg + c) * 32 + g * 8 + kb2 * 4;
                    const f32x4 bv = bq[c][g];
                    const float vv[4] = {acc[cp][g * 4 + 0] + bv[0], acc[cp][g * 4 + 1] + bv[1], acc[cp][g * 4 + 2] + bv[2],
                                         acc[cp][g * 4 + 3] + bv[3]};
                    const int off = row * G::RB + (((ch >> 3) ^ (row & G::SWZ)) << 4) + (ch & 7) * 2;
                    Quad<E> hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                        hi.e[i] = (E)r;
                        lo.e[i] = (E)(r - (float)hi.e[i]);
                    }
                    *reinterpret_cast<Quad<E>*>(Y + off) = hi;
                    if (PARTS == 2) *reinterpret_cast<Quad<E>*>(Y + G::PART_BYTES + off) = lo;
                }
            }
        }
        __syncthreads();                                       // B: Y complete
        __builtin_amdgcn_s_setprio(3);
        conv_kloop<E, C, P, PARTS, CTW>(Y, wq2, lane, acc);
        __builtin_amdgcn_s_setprio(0);
        bias_fetch(b2);
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 2: relu(acc + b2 + x) -> staging
#pragma unroll
        for (int cp = 0; cp < CTW * NT; ++cp) {
            const int p = cp % NT, c = cp / NT;
            const int q = (p % 3) * 32 + ln2;
            const int row = (p / 3) * 90 + q;
            if (q < 90) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = (wg + c) * 32 + g * 8 + kb2 * 4;
                    const f32x4 bv = bq[c][g];
                    const int off = row * G::RB + (((ch >> 3) ^ (row & G::SWZ)) << 4) + (ch & 7) * 2;
                    const Quad<E> sh = *reinterpret_cast<const Quad<E>*>(X + off);
                    float vv[4] = {acc[cp][g * 4 + 0] + bv[0], acc[cp][g * 4 + 1] + bv[1], acc[cp][g * 4 + 2] + bv[2],
                                   acc[cp][g * 4 + 3] + bv[3]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) vv[i] += (float)sh.e[i];
                    if (PARTS == 2) {
                        const Quad<E> sl = *reinterpret_cast<const Quad<E>*>(X + G::PART_BYTES + off);
#pragma unroll
                        for (int i = 0; i < 4; ++i) vv[i] += (float)sl.e[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) vv[i] = vv[i] > 0.0f ? vv[i] : 0.0f;
                    if (PARTS == 2) {
                        *reinterpret_cast<float4*>(S + stage_off(row, ch)) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    } else {
                        Quad<E> o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o.e[i] = (E)vv[i];
                        *reinterpret_cast<Quad<E>*>(S + stage_off(row, ch)) = o;
                    }
                }
            }
        }
        __syncthreads();                                       // C: staged; X may be replaced
        if (!has_next) break;
        t += stride;
    }
}

// ---- kernel 2p: a CHAIN of residual blocks on plain 2-byte operands (256 filters: the `deep` 20 x 256 fp16 tower) ---------------
// k_resblock<E, 256, 1, 1, false, 2> keeps X | Y | a staging image and sends every block's result through the copy waves to HBM and
// back.  With one board per workgroup and plain operands the second epilogue can write its result IN PLACE over the skip operand
// (a lane reads and writes only its own 8 bytes), which is block b + 1's input: a workgroup takes a board through ALL blocks of
// the tower in X | Y, HBM sees it at the entry and the exit, the matrix waves never wait for a refill between blocks.  Same K loop
// (conv_kloop, two channel tiles per wave) and the same epilogue arithmetic in the same order: bit-identical to ch.n launches of
// k_resblock (BASELINE configs[4]; reference tower agent/model.py:41-43).
namespace pl {
constexpr int MAX_BLOCKS = 24;
struct Chain {
    const void* w1[MAX_BLOCKS];
    const void* w2[MAX_BLOCKS];
    const float* b1[MAX_BLOCKS];
    const float* b2[MAX_BLOCKS];
    int n;
};
}  // namespace pl

template <typename E, int C, int CTW>
__global__ __launch_bounds__((C / 32 / CTW + 4) * 64, (C / 32 / CTW + 4 + 3) / 4) void k_tower_plain(
    const E* __restrict__ xh, pl::Chain ch, E* __restrict__ yh, int n_boards, const int32_t* __restrict__ n_dev)
{
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    typedef Geom<C, 1, 1> G;
    constexpr int NT = G::NT, CT = G::CT / CTW, CTHR = RB_COPY_THREADS;
    constexpr int LITER = (G::CHUNKS + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G::REGION];
    unsigned char* X = lds;
    unsigned char* Y = lds + G::REGION;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NB = ch.n;
    int t = blockIdx.x;
    if (t >= n_boards) return;
    const int stride = gridDim.x;

    if (wave >= CT) {                                   // ---- copy waves ----
        const int ctid = tid - CT * 64;
        uint4 v[1][LITER];
        tile_load<E, C, 1, 1, CTHR>(xh, nullptr, t, n_boards, ctid, v);
        tile_write<C, 1, 1, CTHR>(X, ctid, v);
        zero_rows_write<C, 1, 1, CTHR>(X, ctid);
        zero_rows_write<C, 1, 1, CTHR>(Y, ctid);
        for (;;) {
            __syncthreads();                                   // A: X holds board t
            const int tn = t + stride;
            const bool has_next = tn < n_boards;
            int ct2 = ctid;
            asm volatile("" : "+v"(ct2));
            if (has_next) tile_load<E, C, 1, 1, CTHR>(xh, nullptr, tn, n_boards, ct2, v);
            for (int b = 0; b < NB; ++b) {
                __syncthreads();                               // B
                __syncthreads();                               // C: block b's result is in X
            }
            // the chain's result to HBM; the same chunks take the next board
            uint4* dst = reinterpret_cast<uint4*>(yh + (size_t)t * 90 * C);
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int i = it * CTHR + ct2;
                if (!((it + 1) * CTHR <= G::CHUNKS || i < G::CHUNKS)) continue;
                const int row = i / G::CPR, chn = i % G::CPR;
                unsigned char* a = X + row * G::RB + ((chn ^ (row & G::SWZ)) << 4);
                const uint4 o = *reinterpret_cast<const uint4*>(a);
                if (has_next) *reinterpret_cast<uint4*>(a) = v[0][it];
                dst[i] = o;
            }
            if (!has_next) break;
            t = tn;
        }
        return;
    }

    // ---- matrix waves ----
    const int wg = wave * CTW;                                 // first channel tile of this wave
    const int kb = lane >> 5, ln = lane & 31;
    for (;;) {
        __syncthreads();                                       // A
        const bool has_next = t + stride < n_boards;
        for (int blk = 0; blk < NB; ++blk) {
            const uint4* wq1 = reinterpret_cast<const uint4*>(ch.w1[blk]) + wg * 64 + lane;
            const uint4* wq2 = reinterpret_cast<const uint4*>(ch.w2[blk]) + wg * 64 + lane;
            const float* b1 = ch.b1[blk];
            const float* b2 = ch.b2[blk];
            f32x16 acc[CTW * NT];
            __builtin_amdgcn_s_setprio(3);
            conv_kloop<E, C, 1, 1, CTW>(X, wq1, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            f32x4 bq[CTW][4];                                  // (k_resblock's bias_fetch)
            auto bias_fetch = [&](const float* bp) {
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) bq[c][g] = *reinterpret_cast<const f32x4*>(bp + (wg + c) * 32 + g * 8 + kb * 4);
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[c][g]));
            };
            bias_fetch(b1);
            int ln2 = ln, kb2 = kb;
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 1: relu(acc + b1) -> Y (k_resblock's)
#pragma unroll
            for (int cp = 0; cp < CTW * NT; ++cp) {
                const int p = cp % NT, c = cp / NT;
                const int q = p * 32 + ln2;
                if (q < 90) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chn = (wg + c) * 32 + g * 8 + kb2 * 4;
                        const f32x4 bv = bq[c][g];
                        const float vv[4] = {acc[cp][g * 4 + 0] + bv[0], acc[cp][g * 4 + 1] + bv[1], acc[cp][g * 4 + 2] + bv[2],
                                             acc[cp][g * 4 + 3] + bv[3]};
                        const int off = q * G::RB + (((chn >> 3) ^ (q & G::SWZ)) << 4) + (chn & 7) * 2;
                        Quad<E> hi;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                            hi.e[i] = (E)r;
                        }
                        *reinterpret_cast<Quad<E>*>(Y + off) = hi;
                    }
                }
            }
            __syncthreads();                                   // B: Y complete
            __builtin_amdgcn_s_setprio(3);
            conv_kloop<E, C, 1, 1, CTW>(Y, wq2, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            bias_fetch(b2);
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 2: relu(acc + b2 + x) -> X, in place over the skip operand (this lane's own 8 bytes)
#pragma unroll
            for (int cp = 0; cp < CTW * NT; ++cp) {
                const int p = cp % NT, c = cp / NT;
                const int q = p * 32 + ln2;
                if (q < 90) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chn = (wg + c) * 32 + g * 8 + kb2 * 4;
                        const f32x4 bv = bq[c][g];
                        const int off = q * G::RB + (((chn >> 3) ^ (q & G::SWZ)) << 4) + (chn & 7) * 2;
                        const Quad<E> sh = *reinterpret_cast<const Quad<E>*>(X + off);
                        float vv[4] = {acc[cp][g * 4 + 0] + bv[0], acc[cp][g * 4 + 1] + bv[1], acc[cp][g * 4 + 2] + bv[2],
                                       acc[cp][g * 4 + 3] + bv[3]};
                        Quad<E> o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            vv[i] += (float)sh.e[i];
                            vv[i] = vv[i] > 0.0f ? vv[i] : 0.0f;
                            o.e[i] = (E)vv[i];
                        }
                        *reinterpret_cast<Quad<E>*>(X + off) = o;
                    }
                }
            }
            __syncthreads();                                   // C: the block's result is in X
        }
        if (!has_next) break;
        t += stride;
    }
}

// ---- kernel 2pp: the same chain for a PAIR of boards per workgroup, ONE LDS image per board -------------------------------------
// What bounds k_tower_plain is energy per board (EXPERIMENTS Part III): per board and convolution 1.18 MB of packed filter come
// from L2 into the CU.  Here a filter fragment feeds SIX pixel tiles (two boards) instead of three: half the filter stream per
// board, same LDS reads per MFMA.  Two boards with X | Y each do not fit 160 KB; they do with ONE image per board: after K loop 1
// (barrier: every wave has read all of X) a lane moves its own skip values from X into registers (96 values per board) and
// writes relu(conv1 + b1) over them -- X now IS Y --, and epilogue 2 writes relu(conv2 + b2 + skip) to the same 8 bytes again.
// 192 accumulator + 96 skip registers per wave: four matrix waves, no copy waves, one wave per SIMD (512 registers); the matrix
// waves fetch and store the pair themselves, once per chain.  K loop (conv_kloop<E, 256, 2, 1, 2>) and epilogue arithmetic are
// k_resblock's, per accumulator tile in the same order: bit-identical to k_tower_plain and to ch.n launches of k_resblock.
#ifndef CZ_TP2_BIAS_EARLY
#define CZ_TP2_BIAS_EARLY 1
#endif
#ifndef CZ_TP2_TILE_FENCE
#define CZ_TP2_TILE_FENCE 0
#endif
#ifndef CZ_TP2_SKIP_LDS
#define CZ_TP2_SKIP_LDS 1
#endif
template <typename E, int C, int CTW>
__global__ __launch_bounds__(C / 32 / CTW * 64, 1) void k_tower_plain2(
    const E* __restrict__ xh, pl::Chain ch, E* __restrict__ yh, int n_boards, const int32_t* __restrict__ n_dev)
{
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    typedef Geom<C, 2, 1> G;
    constexpr int NT = G::NT, NTHR = C / 32 / CTW * 64;
    constexpr int LITER = (G::CHUNKS + NTHR - 1) / NTHR;
    // X: the pair's image.  SK: the FIRST board's skip values while its intermediate activation sits in X (the second board's wait
    // in registers: 48 -- all 96 in registers left no room beside 192 accumulators and the K loop's rings: half of them spilled)
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::REGION + (CZ_TP2_SKIP_LDS ? 90 * G::RB : 0)];
    unsigned char* X = lds;
    unsigned char* SK = lds + G::REGION;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NB = ch.n;
    const int n_pairs = (n_boards + 1) / 2;
    const int wg = wave * CTW;                                 // first channel tile of this wave
    const int kb = lane >> 5, ln = lane & 31;
    zero_rows_write<C, 2, 1, NTHR>(X, tid);
    for (int t = blockIdx.x; t < n_pairs; t += gridDim.x) {
        {
            // global -> LDS image, 16 bytes per lane per step in flights of eight (a missing second board: zeros)
            typedef c8k::u32x4 u4;
            const u4* src = reinterpret_cast<const u4*>(xh + (size_t)2 * t * 90 * C);
            const int have = (n_boards - 2 * t < 2 ? n_boards - 2 * t : 2) * 90 * G::CPR;
#pragma unroll
            for (int it0 = 0; it0 < LITER; it0 += 8) {
                u4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = (it0 + j) * NTHR + tid;
                    v[j] = u4{0u, 0u, 0u, 0u};
                    if (it0 + j < LITER && i < have) v[j] = src[i];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = (it0 + j) * NTHR + tid;
                    if (it0 + j < LITER && i < G::CHUNKS) {
                        const int row = i / G::CPR, chn = i % G::CPR;
                        *reinterpret_cast<u4*>(X + row * G::RB + ((chn ^ (row & G::SWZ)) << 4)) = v[j];
                    }
                }
            }
        }
        __syncthreads();                                       // A: X holds the pair
        for (int blk = 0; blk < NB; ++blk) {
            const uint4* wq1 = reinterpret_cast<const uint4*>(ch.w1[blk]) + wg * 64 + lane;
            const uint4* wq2 = reinterpret_cast<const uint4*>(ch.w2[blk]) + wg * 64 + lane;
            const float* b1 = ch.b1[blk];
            const float* b2 = ch.b2[blk];
            f32x16 acc[CTW * NT];
            c8k::u32x2 skip[CZ_TP2_SKIP_LDS ? CTW * 3 : CTW * NT][4];          // (packed: four 2-byte values in two registers)
            f32x4 bq[CTW][4];
            // the biases of this wave's channels: all eight loads in flight across the barrier, materialised once (left to the
            // compiler there was one load and one vmcnt(0) per 8-byte store: 48 L2 round trips in a row per epilogue)
            auto bias_request = [&](const float* bp) {
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) bq[c][g] = *reinterpret_cast<const f32x4*>(bp + (wg + c) * 32 + g * 8 + kb * 4);
            };
            auto bias_pin = [&]() {
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[c][g]));
            };
            __builtin_amdgcn_s_setprio(3);
            conv_kloop<E, C, 2, 1, CTW>(X, wq1, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            if (CZ_TP2_BIAS_EARLY) bias_request(b1);
            __syncthreads();                                   // K1: every wave has read X
            if (CZ_TP2_BIAS_EARLY) bias_pin();
            int ln2 = ln, kb2 = kb;
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 1: skip <- X, X <- relu(acc + b1)   (this lane's own 8 bytes)
#pragma unroll
            for (int cp = 0; cp < CTW * NT; ++cp) {
                const int p = cp % NT, c = cp / NT;
                const int q = (p % 3) * 32 + ln2;
                const int row = (p / 3) * 90 + q;
                if (q < 90) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chn = (wg + c) * 32 + g * 8 + kb2 * 4;
                        const f32x4 bv = CZ_TP2_BIAS_EARLY ? bq[c][g] : *reinterpret_cast<const f32x4*>(b1 + chn);
                        const float vv[4] = {acc[cp][g * 4 + 0] + bv[0], acc[cp][g * 4 + 1] + bv[1], acc[cp][g * 4 + 2] + bv[2],
                                             acc[cp][g * 4 + 3] + bv[3]};
                        const int off = row * G::RB + (((chn >> 3) ^ (row & G::SWZ)) << 4) + (chn & 7) * 2;
                        const c8k::u32x2 sk = *reinterpret_cast<const c8k::u32x2*>(X + off);
                        if (!CZ_TP2_SKIP_LDS) skip[cp][g] = sk;
                        else if (p < 3) *reinterpret_cast<c8k::u32x2*>(SK + off) = sk;
                        else skip[c * 3 + p - 3][g] = sk;
                        Quad<E> hi;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                            hi.e[i] = (E)r;
                        }
                        *reinterpret_cast<Quad<E>*>(X + off) = hi;
                    }
                }
                if (CZ_TP2_TILE_FENCE) __builtin_amdgcn_sched_barrier(0);     // one accumulator tile at a time (they leave the AGPRs 16 by 16)
            }
            __syncthreads();                                   // B: X holds the intermediate activation
            __builtin_amdgcn_s_setprio(3);
            conv_kloop<E, C, 2, 1, CTW>(X, wq2, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            if (CZ_TP2_BIAS_EARLY) bias_request(b2);
            __syncthreads();                                   // K2: every wave has read it
            if (CZ_TP2_BIAS_EARLY) bias_pin();
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 2: X <- relu(acc + b2 + skip)
#pragma unroll
            for (int cp = 0; cp < CTW * NT; ++cp) {
                const int p = cp % NT, c = cp / NT;
                const int q = (p % 3) * 32 + ln2;
                const int row = (p / 3) * 90 + q;
                if (q < 90) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chn = (wg + c) * 32 + g * 8 + kb2 * 4;
                        const f32x4 bv = CZ_TP2_BIAS_EARLY ? bq[c][g] : *reinterpret_cast<const f32x4*>(b2 + chn);
                        const int off = row * G::RB + (((chn >> 3) ^ (row & G::SWZ)) << 4) + (chn & 7) * 2;
                        float vv[4] = {acc[cp][g * 4 + 0] + bv[0], acc[cp][g * 4 + 1] + bv[1], acc[cp][g * 4 + 2] + bv[2],
                                       acc[cp][g * 4 + 3] + bv[3]};
                        const c8k::u32x2 sk = !CZ_TP2_SKIP_LDS ? skip[cp][g]
                                              : (p < 3 ? *reinterpret_cast<const c8k::u32x2*>(SK + off) : skip[c * 3 + p - 3][g]);
                        const Quad<E> sh = __builtin_bit_cast(Quad<E>, sk);
                        Quad<E> o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            vv[i] += (float)sh.e[i];
                            vv[i] = vv[i] > 0.0f ? vv[i] : 0.0f;
                            o.e[i] = (E)vv[i];
                        }
                        *reinterpret_cast<Quad<E>*>(X + off) = o;
                    }
                }
                if (CZ_TP2_TILE_FENCE) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                   // C: the block's result is in X
        }
        // the chain's result to HBM
        const int valid = (n_boards - 2 * t < 2 ? n_boards - 2 * t : 2) * 90 * G::CPR;
        c8k::u32x4* dst = reinterpret_cast<c8k::u32x4*>(yh + (size_t)2 * t * 90 * C);
#pragma unroll
        for (int it = 0; it < LITER; ++it) {
            const int i = it * NTHR + tid;
            if (!((it + 1) * NTHR <= G::CHUNKS || i < G::CHUNKS) || i >= valid) continue;
            const int row = i / G::CPR, chn = i % G::CPR;
            dst[i] = *reinterpret_cast<const c8k::u32x4*>(X + row * G::RB + ((chn ^ (row & G::SWZ)) << 4));
        }
        __syncthreads();                                       // X may take the next pair
    }
}

#ifdef CZ_RB_STAMPS
// timing build (tools/rb_stamps.py; never the default library): shader-cycle stamps of one matrix wave's and one copy
// wave's sections of one steady-state board of k_resblock_c8
__device__ long long g_rb_stamps[32];
__device__ int g_rb_knob;              // timing experiments (wrong results): bits switch parts of the c6 store path off
#define RB_STAMP(i) do { if (stamp_on) g_rb_stamps[i] = clock64(); } while (0)
#define RB_KNOB() g_rb_knob
#else
#define RB_KNOB() 0
#define RB_STAMP(i) do { } while (0)
#endif

// ---- kernel 2c8: the residual block on the c8 arithmetic (round 4) -------------------------------------------------------
// k_resblock's roles and LDS images (X | Y | fp32 staging; 4 matrix waves + 4 copy waves, one workgroup per CU), with the
// critical path of a board cut down to what in-kernel cycle stamps showed it to be (tools/rb_stamps.py on round 3's
// schedule: 44.2 k shader cycles per board = two K loops 20.0 + 16.2 k, epilogue 1 3.8 k, epilogue 2 2.8 k, 1.2 k waiting for
// the copy waves to hand X over):
//   * accumulators START at bias (K loop 1) and at bias + skip (K loop 2; read from X by the lane that owns the element,
//     unit by unit as epilogue 1 frees the registers): no additions left in the epilogues, and X is dead once K loop 2 starts;
//   * epilogue 2 is only  relu -> fp32 staging  and is DEFERRED into the fp8 slots of the next board's first K loop
//     (12 ds_write_b128 per wave, one every ninth slot; c8k::kloop's shadow hook) -- the conversion to the c8 triple stays
//     with the copy waves, whose instructions do not compete with the matrix waves' issue slots the way shadow VALU work
//     does (the fully in-place pipelined form, 45 VALU per unit in the shadow, measured 7 % SLOWER than the plain one);
//   * the copy waves write the next board into X during K loop 2 (X is free from barrier B on) and store the previous
//     board from the staging image there too; under K loop 1 they only issue the next board's loads.  Two barriers per
//     board instead of three, none of them waited at by the matrix waves.
//   matrix waves, board k:  A | K1(k) + shadow: staging <- relu(acc2(k-1)) | epi1 -> Y, acc <- b2 + skip | B | K2(k) | ...
//   copy waves:             A | loads of board k+1 -> registers            |                            | B | store board
//                               k-1 from staging (c8 split, or fp32, or the head convolutions); X <- board k+1
// FIRST: the block is the first of the tower; its copy waves compute the 5 x 5 input layer of board k+1 as an fp32 gather
// over the occupied squares (see k_resblock_pipe<FIRST> below: the same algorithm, here producing the c8 triple) instead
// of loading it.  HEADS: the last block; the copy waves apply the two 1 x 1 head convolutions to the staged activation.
// Arithmetic and its order are k_conv3x3_c8's: bit-identical to two cz_conv3x3_c8 launches.
#ifndef CZ_FIRST_TERMS
#define CZ_FIRST_TERMS 1
#endif
constexpr int CZ_FIRST_PRIO = 3;   // issue priority of the copy waves while they compute the fused input layer (the matrix waves
                                   // run their K loops at 3; at 0 the gather starves: first block 3.53 -> 3.48 ms)
namespace rb8 {
constexpr int C = 128, ZROW = 96, PART = (ZROW + 16) * RB, REGION = 2 * PART;
constexpr int SROW = 512, S_BYTES = 90 * SROW;
constexpr int Y_OFF = REGION, S_OFF = 2 * REGION, BIAS_OFF = S_OFF + S_BYTES, MASK_OFF = BIAS_OFF + 2 * C * 4;
constexpr int LDS_BYTES = MASK_OFF + 4 * 96 * 4;
constexpr int DUMP_OFF = Y_OFF + 90 * RB;      // rows 90 .. 95 of Y are never read: the shadow's padding lanes write here
static_assert(LDS_BYTES <= 160 * 1024, "X + Y + staging + bias + mask boards must fit the CU's LDS");
// staging offset of channels ch .. ch + 3 of pixel q (fp32, 32 chunks of 16 bytes per row, swizzled by the row)
__device__ __forceinline__ int stage_off(int q, int ch) { return q * SROW + ((((ch >> 2) & ~7) | (((ch >> 2) ^ q) & 7)) << 4); }

struct Shadow {                         // relu(acc2 of the previous board) -> staging, one (tile, channel group) unit per 9 fp8 slots
    unsigned char* lds;
    f32x16* prev;                       // the body always reads prev[0]; rotated at the end of an iteration
    int wave, kb, ln;
    __device__ __forceinline__ void unit(const f32x16& a, int q, int g)
    {
        const int ch = wave * 32 + g * 8 + kb * 4;
        const int addr = q < 90 ? S_OFF + stage_off(q, ch) : DUMP_OFF + (kb * 6 + (ln - 26)) * 16;
        float4 v;
        v.x = a[g * 4 + 0] > 0.0f ? a[g * 4 + 0] : 0.0f;
        v.y = a[g * 4 + 1] > 0.0f ? a[g * 4 + 1] : 0.0f;
        v.z = a[g * 4 + 2] > 0.0f ? a[g * 4 + 2] : 0.0f;
        v.w = a[g * 4 + 3] > 0.0f ? a[g * 4 + 3] : 0.0f;
        *reinterpret_cast<float4*>(lds + addr) = v;
    }
    __device__ __forceinline__ void fp8(int j, int slot)
    {
        if (slot % 9 == 4) unit(prev[0], j * 32 + ln, slot / 9);
        if (slot == 35) {
            prev[0] = prev[1];
            prev[1] = prev[2];
        }
    }
};
}  // namespace rb8

template <bool FIRST, bool HEADS, bool C6 = false>
__global__ __launch_bounds__(512, 2) void k_resblock_c8(
    const _Float16* __restrict__ xh, const unsigned char* __restrict__ xc, const void* __restrict__ w1p,
    const float* __restrict__ b1, const void* __restrict__ w2p, const float* __restrict__ b2, _Float16* __restrict__ yh,
    unsigned char* __restrict__ yc, float* __restrict__ yf, int n_boards, HeadArgs hd, const int32_t* __restrict__ n_dev,
    FirstArgs fa)
{
    using namespace rb8;
    static_assert(!(FIRST && HEADS), "a one-block tower runs cz_input_conv + the HEADS block");
    typedef Geom<C, 1, 2> G;
    static_assert(G::ZROW == ZROW && G::PART_BYTES == PART && G::REGION == REGION, "LDS image geometry");
    constexpr int NT = 3, CTHR = 256, CHUNKS = 90 * 16, LITER = (CHUNKS + CTHR - 1) / CTHR;
    constexpr int PIECES = 90 * (C / 8), EITER = (PIECES + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    if (n_dev) {                                        // compact queue: the board count lives on the device
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    unsigned char* X = lds;
    unsigned char* Y = lds + Y_OFF;
    unsigned char* S = lds + S_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = gridDim.x;
    int t = blockIdx.x;
    if (t >= n_boards) return;

    if (wave >= 4) {                                    // ---- copy waves ----
        const int ctid = tid - 256;
        uint4 v[2][LITER];
        float hw[HEADS ? 6 : 1][8];                     // this thread's slice of the head filters
        if (HEADS) {
#pragma unroll
            for (int o = 0; o < 6; ++o)
#pragma unroll
                for (int k = 0; k < 8; ++k) hw[o][k] = hd.w[o * C + (ctid % (C / 8)) * 8 + k];
        }
        // board `to` from the staging image (fp32, already ReLU'd) to HBM
        auto store_tile = [&](int to, int ct2) __attribute__((always_inline)) {
            const size_t ebase = (size_t)to * 90 * C;
#pragma unroll
            for (int it = 0; it < EITER; ++it) {
                const int i = it * CTHR + ct2;
                if (!((it + 1) * CTHR <= PIECES || i < PIECES)) continue;
                const int qq = i / (C / 8), c8 = i % (C / 8);
                const float4 f0 = *reinterpret_cast<const float4*>(S + stage_off(qq, c8 * 8));
                const float4 f1 = *reinterpret_cast<const float4*>(S + stage_off(qq, c8 * 8 + 4));
                const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                if (HEADS) {
                    // 16 consecutive lanes hold the 128 channels of one pixel: partial dot products, then a 16-lane
                    // butterfly; lane o of the group writes head output o
                    float hs[6];
#pragma unroll
                    for (int o = 0; o < 6; ++o) {
                        float a = 0.0f;
#pragma unroll
                        for (int k = 0; k < 8; ++k) a += r[k] * hw[o][k];
                        a = row16_sum(a);
                        hs[o] = a;
                    }
#pragma unroll
                    for (int o = 0; o < 6; ++o)
                        if (c8 == o) {
                            float hv = hs[o] + hd.b[o];
                            hv = hv > 0.0f ? hv : 0.0f;
                            if (o < hd.n_pol) hd.pol[(size_t)to * (hd.n_pol * 90) + o * 90 + qq] = hv;
                            else hd.val[(size_t)to * ((6 - hd.n_pol) * 90) + (o - hd.n_pol) * 90 + qq] = hv;
                        }
                } else if (yf) {
                    float4* o = reinterpret_cast<float4*>(yf + ebase) + 2 * i;
                    o[0] = f0;
                    o[1] = f1;
                } else {
                    const cf8::Split4 s0 = cf8::split4(r), s1 = cf8::split4(r + 4);
                    struct alignas(16) H8 { Quad<_Float16> a, b; };
                    reinterpret_cast<uint4*>(yh + ebase)[i] = __builtin_bit_cast(uint4, H8{s0.hi, s1.hi});
                    unsigned char* row = yc + ebase * 2 + (size_t)qq * 2 * C + c8 * 8;
                    *reinterpret_cast<uint2*>(row) = make_uint2(s0.l8, s1.l8);
                    *reinterpret_cast<uint2*>(row + C) = make_uint2(s0.h8, s1.h8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // the same for a c6 output image.  Pass A: the f16 halves in store_tile's mapping (a thread = 8 channels of a pixel:
        // 16 bytes, consecutive over the lanes -- whole cache lines per instruction).  Pass B: one thread takes the 32 channels
        // of a (pixel, block) -- a whole piece of each kind (24 bytes, 4 neighbouring lanes fill 96 of a line's 128).
        // (The K loops' filter stream keeps the CU's vector-memory path ~80 % busy -- 576 KB per board and convolution at
        //  64 B/clk --, so every extra memory instruction of the copy waves shows: an earlier form that wrote the f16 halves
        //  from pass B's mapping, 16 bytes at a 64-byte lane stride, cost 0.1 ms per launch more; streaming (nt) stores 0.13.)
        const int k_out = C6 ? __builtin_amdgcn_readfirstlane(pack_ints(w2p)[3]) : 0;
        auto store_tile_c6 = [&](int to, int ct2) __attribute__((always_inline)) {
            const int knob = RB_KNOB();
            if (knob & 4) return;
            const size_t ebase = (size_t)to * 90 * C;
            const float s_hi = __builtin_ldexpf(1.0f, k_out), s_lo = __builtin_ldexpf(1.0f, k_out - cf8::X_LO_SHIFT);
            struct alignas(16) H8 { Quad<_Float16> a, b; };
#pragma unroll
            for (int it = 0; it < EITER; ++it) {               // ---- pass A
                const int i = it * CTHR + ct2;
                if (!((it + 1) * CTHR <= PIECES || i < PIECES)) continue;
                const int qq = i / (C / 8), c8 = i % (C / 8);
                const float4 f0 = *reinterpret_cast<const float4*>(S + stage_off(qq, c8 * 8));
                const float4 f1 = *reinterpret_cast<const float4*>(S + stage_off(qq, c8 * 8 + 4));
                H8 h;
                h.a.e[0] = (_Float16)f0.x; h.a.e[1] = (_Float16)f0.y; h.a.e[2] = (_Float16)f0.z; h.a.e[3] = (_Float16)f0.w;
                h.b.e[0] = (_Float16)f1.x; h.b.e[1] = (_Float16)f1.y; h.b.e[2] = (_Float16)f1.z; h.b.e[3] = (_Float16)f1.w;
                if (!(knob & 1)) reinterpret_cast<uint4*>(yh + ebase)[i] = __builtin_bit_cast(uint4, h);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {                   // ---- pass B
                const int i = it * CTHR + ct2;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, blk = i & 3;
                f32x16 av, bv, al, bl;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 f0 = *reinterpret_cast<const float4*>(S + stage_off(qq, blk * 32 + 8 * k));
                    const float4 f1 = *reinterpret_cast<const float4*>(S + stage_off(qq, blk * 32 + 8 * k + 4));
                    const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        av[4 * k + j] = r[j];
                        bv[4 * k + j] = r[4 + j];
                        al[4 * k + j] = r[j] - (float)(_Float16)r[j];
                        bl[4 * k + j] = r[4 + j] - (float)(_Float16)r[4 + j];
                    }
                }
                const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
                const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
                unsigned char* row = yc + ebase * 2 + (size_t)qq * 2 * C;
                if (knob & 2) continue;
                *reinterpret_cast<uint4*>(row + 16 * c6_chunk(0, blk)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                *reinterpret_cast<uint2*>(row + 16 * c6_chunk(0, blk) + 16 + c6_tail_half(qq)) = make_uint2(pl[4], pl[5]);
                *reinterpret_cast<uint4*>(row + 16 * c6_chunk(1, blk)) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                *reinterpret_cast<uint2*>(row + 16 * c6_chunk(1, blk) + 16 + c6_tail_half(qq)) = make_uint2(pv[4], pv[5]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // (two call sites each; a wrapper lambda around them was NOT inlined by the compiler: a call inside the kernel, 2.5x
        //  the scratch and 17 % of the launch time)
        // (k_out == CZ_C6_OUT_C8: the last c6 block of a hybrid c6>N tower hands a c8 image to the c8 blocks behind it)
#define CZ_STORE_BOARD(to, ct2) do { if (C6 && !HEADS && !yf && k_out != CZ_C6_OUT_C8 && !(RB_KNOB() & 8)) store_tile_c6(to, ct2); else store_tile(to, ct2); } while (0)
        // registers -> X image.  Loaded boards: 16-byte chunks of both parts; FIRST: the thread's 8 channels of a pixel are
        // one 16-byte chunk of the f16 row and two 8-byte pieces of the c8 row [lo8 x 128 | e4m3(x) x 128]
        auto write_x = [&](int ct2) {
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int i = it * CTHR + ct2;
                if (!((it + 1) * CTHR <= CHUNKS || i < CHUNKS)) continue;
                const int row = i >> 4, ch = i & 15;
                *reinterpret_cast<uint4*>(X + row * RB + ((ch ^ (row & 15)) << 4)) = v[0][it];
                if (FIRST) {
                    unsigned char* r = X + PART + row * RB + (ch & 1) * 8;
                    *reinterpret_cast<uint2*>(r + (((ch >> 1) ^ (row & 15)) << 4)) = make_uint2(v[1][it].x, v[1][it].y);
                    *reinterpret_cast<uint2*>(r + (((8 + (ch >> 1)) ^ (row & 15)) << 4)) = make_uint2(v[1][it].z, v[1][it].w);
                } else {
                    *reinterpret_cast<uint4*>(X + PART + row * RB + ((ch ^ (row & 15)) << 4)) = v[1][it];
                }
            }
        };
        // ---- FIRST: the input layer of a board into v[][], in two halves (k_resblock_pipe<FIRST> documents the algorithm)
        float acc[FIRST ? LITER : 1][8];
        uint32_t occ[FIRST ? LITER : 1], cur_m[FIRST ? LITER : 1];
        int cur_tap[FIRST ? LITER : 1];
        uint32_t* mk = reinterpret_cast<uint32_t*>(lds + MASK_OFF) + (wave - 4) * 96;     // this wave's own mask board
        const int c8 = ctid & 15, prow = ctid >> 4;            // chunk = channels 8 c8 .. 8 c8 + 7 of pixels prow + 16 it
        constexpr int PW = 12;                                 // plane words per lane: 32 planes x 90 bytes / 4 / 64 lanes
        uint32_t pw[FIRST ? PW : 1];
        auto planes_prefetch = [&](int board) {
            const int per_board = fa.in_planes * 90;
            const size_t slot = (size_t)(fa.rows ? fa.rows[board] : board);
            if (fa.masks) {                                    // the board's occupancy board, as the search kernel wrote it
                pw[0] = fa.masks[slot * 96 + lane];
                pw[1] = lane < 32 ? fa.masks[slot * 96 + 64 + lane] : 0u;
                return;
            }
            const uint32_t* src = reinterpret_cast<const uint32_t*>(fa.planes + slot * per_board);
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                const int w = lane + 64 * j;
                pw[j] = w < per_board / 4 ? src[w] : 0u;
            }
        };
        auto first_begin = [&]() {
            if (fa.masks) {                                    // (round 5) the mask board arrives ready-made
                mk[lane] = pw[0];
                if (lane < 32) mk[64 + lane] = pw[1];
            } else {
                mk[lane] = 0u;
                if (lane < 32) mk[64 + lane] = 0u;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    const int w = lane + 64 * j;
                    const uint32_t word = pw[j];
                    if (word == 0u) continue;
                    int c = (w * 4) / 90, pix = w * 4 - c * 90;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        if ((word >> (8 * k4)) & 0xFFu) atomicOr(&mk[pix], 1u << c);
                        if (++pix == 90) { pix = 0; ++c; }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const uint64_t occ_lo = __ballot(mk[lane] != 0u);
            const uint64_t occ_hi = __ballot(lane < 26 && mk[64 + (lane & 31)] != 0u);
            const float4 b0 = *reinterpret_cast<const float4*>(fa.in_bias + c8 * 8);
            const float4 b1v = *reinterpret_cast<const float4*>(fa.in_bias + c8 * 8 + 4);
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                acc[it][0] = b0.x; acc[it][1] = b0.y; acc[it][2] = b0.z; acc[it][3] = b0.w;
                acc[it][4] = b1v.x; acc[it][5] = b1v.y; acc[it][6] = b1v.z; acc[it][7] = b1v.w;
                const int p = prow + 16 * it;
                const int py = p / 9, px = p - py * 9;
                uint32_t o = 0u;
                if (p < 90) {
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) {
                        const int r = py + ky - 2;
                        if ((unsigned)r < 10u) {
                            const int sh = r * 9;
                            const uint32_t rowbits = (uint32_t)((sh < 64 ? (occ_lo >> sh) | (sh > 55 ? occ_hi << (64 - sh) : 0ull)
                                                                         : occ_hi >> (sh - 64)) & 0x1FFull);
                            o |= (((rowbits << 2) >> px) & 31u) << (5 * ky);
                        }
                    }
                }
                occ[it] = o;
                cur_m[it] = 0u;
                cur_tap[it] = 0;
            }
        };
        // CZ_FIRST_TERMS (build switch, 1 or 2): table rows fetched per chunk and ROUND.  A round is one L2 round trip; with 2 the
        // walk of a board takes half as many of them (two rows in flight per chunk) at 48 more registers; the rows are added
        // in the same order either way (bit-identical results).  max_rounds counts TERMS.
#if CZ_FIRST_TERMS == 1
        auto first_rounds = [&](int max_rounds) {
#pragma unroll 1
            for (int round = 0; round < max_rounds; ++round) {
                uint32_t any = 0u;
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    if (cur_m[it] == 0u && occ[it] != 0u) {
                        const int tap = __builtin_ctz(occ[it]);
                        occ[it] &= occ[it] - 1u;
                        const int p = prow + 16 * it;
                        cur_tap[it] = tap;
                        cur_m[it] = mk[p + (tap / 5 - 2) * 9 + (tap % 5 - 2)];
                    }
                    any |= cur_m[it];
                }
                if (!__ballot(any != 0u)) break;
                float4 wa[LITER], wb[LITER];
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    wa[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    wb[it] = wa[it];
                    if (cur_m[it]) {
                        const int c = __builtin_ctz(cur_m[it]);
                        const float4* tp = reinterpret_cast<const float4*>(fa.table + ((size_t)(c * 25 + cur_tap[it]) * 128 + c8 * 8));
                        wa[it] = tp[0];
                        wb[it] = tp[1];
                    }
                }
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    if (cur_m[it]) {
                        acc[it][0] += wa[it].x; acc[it][1] += wa[it].y; acc[it][2] += wa[it].z; acc[it][3] += wa[it].w;
                        acc[it][4] += wb[it].x; acc[it][5] += wb[it].y; acc[it][6] += wb[it].z; acc[it][7] += wb[it].w;
                        cur_m[it] &= cur_m[it] - 1u;
                    }
                }
            }
        };
#else
        auto first_rounds = [&](int max_rounds) {
            constexpr int FT = CZ_FIRST_TERMS;
#pragma unroll 1
            for (int round = 0; round < max_rounds; round += FT) {
                uint32_t any = 0u;
                bool act[FT][LITER];
                const float4* tp[FT][LITER];
#pragma unroll
                for (int k = 0; k < FT; ++k) {
#pragma unroll
                    for (int it = 0; it < LITER; ++it) {
                        if (cur_m[it] == 0u && occ[it] != 0u) {
                            const int tap = __builtin_ctz(occ[it]);
                            occ[it] &= occ[it] - 1u;
                            const int p = prow + 16 * it;
                            cur_tap[it] = tap;
                            cur_m[it] = mk[p + (tap / 5 - 2) * 9 + (tap % 5 - 2)];
                        }
                        any |= cur_m[it];
                        act[k][it] = cur_m[it] != 0u;
                        tp[k][it] = nullptr;
                        if (act[k][it]) {
                            const int c = __builtin_ctz(cur_m[it]);
                            tp[k][it] = reinterpret_cast<const float4*>(fa.table + ((size_t)(c * 25 + cur_tap[it]) * 128 + c8 * 8));
                            cur_m[it] &= cur_m[it] - 1u;
                        }
                    }
                }
                if (!__ballot(any != 0u)) break;
                float4 wa[FT][LITER], wb[FT][LITER];
#pragma unroll
                for (int k = 0; k < FT; ++k)
#pragma unroll
                    for (int it = 0; it < LITER; ++it) {
                        wa[k][it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        wb[k][it] = wa[k][it];
                        if (act[k][it]) {
                            wa[k][it] = tp[k][it][0];
                            wb[k][it] = tp[k][it][1];
                        }
                    }
#pragma unroll
                for (int k = 0; k < FT; ++k)
#pragma unroll
                    for (int it = 0; it < LITER; ++it) {
                        if (act[k][it]) {
                            acc[it][0] += wa[k][it].x; acc[it][1] += wa[k][it].y; acc[it][2] += wa[k][it].z; acc[it][3] += wa[k][it].w;
                            acc[it][4] += wb[k][it].x; acc[it][5] += wb[k][it].y; acc[it][6] += wb[k][it].z; acc[it][7] += wb[k][it].w;
                        }
                    }
            }
        };
#endif
        auto first_end = [&]() {                              // ReLU, c8 split, pack (see write_x)
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = acc[it][j] > 0.0f ? acc[it][j] : 0.0f;
                const cf8::Split4 s0 = cf8::split4(r), s1 = cf8::split4(r + 4);
                struct alignas(16) H8 { Quad<_Float16> a, b; };
                v[0][it] = __builtin_bit_cast(uint4, H8{s0.hi, s1.hi});
                v[1][it] = make_uint4(s0.l8, s1.l8, s0.h8, s1.h8);
            }
        };
        if (FIRST) {
            planes_prefetch(t);
            first_begin();
            if (t + stride < n_boards) planes_prefetch(t + stride);
            first_rounds(25 * 32);
            first_end();
        } else {
            tile_load<_Float16, C, 1, 2, CTHR>(xh, reinterpret_cast<const _Float16*>(xc), t, n_boards, ctid, v);
        }
        write_x(ctid);
        zero_rows_write<C, 1, 2, CTHR>(X, ctid);
        zero_rows_write<C, 1, 2, CTHR>(Y, ctid);
        if (ctid < C) {
            reinterpret_cast<float*>(lds + BIAS_OFF)[ctid] = b1[ctid];
            reinterpret_cast<float*>(lds + BIAS_OFF)[C + ctid] = b2[ctid];
        }
        int t_prev = -1;
#ifdef CZ_RB_STAMPS
        int it_no = 0;
#endif
        for (;;) {
#ifdef CZ_RB_STAMPS
            const bool stamp_on = blockIdx.x == 5 && ctid == 0 && it_no++ == 40;
#endif
            RB_STAMP(16);
            __syncthreads();                                   // A_k: X holds board t
            RB_STAMP(17);
            const int tn = t + stride;
            const bool has_next = tn < n_boards;
            int ct2 = ctid;
            asm volatile("" : "+v"(ct2));                      // keep address arithmetic inside the loop (registers)
            if (FIRST) {
                if (has_next) {                                // first part of the next board's input layer, under K loop 1
                    __builtin_amdgcn_s_setprio(CZ_FIRST_PRIO);
                    first_begin();                             // (board tn: its planes were fetched a window ago)
                    RB_STAMP(24);
                    if (tn + stride < n_boards) planes_prefetch(tn + stride);
                    RB_STAMP(25);
                    first_rounds(fa.w1_rounds);
                    __builtin_amdgcn_s_setprio(0);
                }
            } else if (has_next) {
                tile_load<_Float16, C, 1, 2, CTHR>(xh, reinterpret_cast<const _Float16*>(xc), tn, n_boards, ct2, v);
            }
            RB_STAMP(18);
            __syncthreads();                                   // B_k: staging holds board t_prev; X is free
            RB_STAMP(19);
            if (t_prev >= 0) CZ_STORE_BOARD(t_prev, ct2);
            RB_STAMP(20);
            if (has_next) {
                if (FIRST) {
                    __builtin_amdgcn_s_setprio(CZ_FIRST_PRIO);
                    first_rounds(25 * 32);
                    RB_STAMP(21);
                    first_end();
                    __builtin_amdgcn_s_setprio(0);
                }
                RB_STAMP(22);
                write_x(ct2);
            }
            RB_STAMP(23);
            t_prev = t;
            if (!has_next) break;
            t = tn;
        }
        __syncthreads();                                       // E0: the staging image is free (the store above is done)
        __syncthreads();                                       // E1: the last board is staged
        CZ_STORE_BOARD(t_prev, ctid);
#undef CZ_STORE_BOARD
        return;
    }

    // ---- matrix waves ----
    const int kb = lane >> 5, ln = lane & 31;
    const c8k::Filter flt1 = c8k::make_filter(w1p, wave, lane), flt2 = c8k::make_filter(w2p, wave, lane);
    const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF);
    const float* bias2 = bias1 + C;
    f32x16 acc[NT], prev[NT];
    // c6: the exponents of the images the two convolutions read (X: not for the fused input layer's c8 image)
    const int k_x = C6 && !FIRST ? __builtin_amdgcn_readfirstlane(pack_ints(w1p)[2]) : 0;
    const int k_y = C6 ? __builtin_amdgcn_readfirstlane(pack_ints(w2p)[2]) : 0;
    rb8::Shadow shd{lds, prev, wave, kb, ln};
    bool have_prev = false;
#ifdef CZ_RB_STAMPS
    int it_no = 0;
#endif
    for (;;) {
#ifdef CZ_RB_STAMPS
        const bool stamp_on = blockIdx.x == 5 && tid == 0 && it_no++ == 40;
#endif
        RB_STAMP(0);
        __syncthreads();                                       // A_k
        RB_STAMP(1);
        const bool has_next = t + stride < n_boards;
        // K loop 1 on X, accumulators starting at b1
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(bias1 + wave * 32 + g * 8 + kb * 4);
#pragma unroll
            for (int p = 0; p < NT; ++p) {
                acc[p][g * 4 + 0] = bv.x; acc[p][g * 4 + 1] = bv.y; acc[p][g * 4 + 2] = bv.z; acc[p][g * 4 + 3] = bv.w;
            }
        }
        __builtin_amdgcn_s_setprio(3);
        constexpr int F1 = C6 && !FIRST ? 1 : 0;              // (the fused input layer hands over a c8 image)
        if (have_prev) c8k::kloop<NT, rb8::Shadow&, 0, false, 128, F1>(X, c8k::Image{0, ZROW, PART}, flt1, lane, acc, 127 + k_x - cf8::X_LO_SHIFT, 127 + k_x, shd);
        else c8k::kloop<NT, c8k::NoShadow, 0, false, 128, F1>(X, c8k::Image{0, ZROW, PART}, flt1, lane, acc, 127 + k_x - cf8::X_LO_SHIFT, 127 + k_x);
        __builtin_amdgcn_s_setprio(0);
        RB_STAMP(2);
        int ln2 = ln, kb2 = kb;
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 1: relu(acc) -> c8 triple -> Y; the freed accumulators restart at b2 + skip (this lane's own elements of X)
        if (C6) {
            // c6: the f16 quads go out per lane as before; the two bf6 pieces of a pixel's 32 channels need the other
            // lane half's 16 values: tiles 0 and 1 trade halves (one v_permlane32_swap per register -- the lower lanes end
            // up with tile 0's pixels, the upper ones with tile 1's), tile 2 trades with itself (lower lanes store)
            const float s_hi = __builtin_ldexpf(1.0f, k_y), s_lo = __builtin_ldexpf(1.0f, k_y - cf8::X_LO_SHIFT);
            const float s_skip = __builtin_ldexpf(1.0f, k_x - cf8::X_LO_SHIFT);
            f32x16 lo[NT];
#pragma unroll
            for (int p = 0; p < NT; ++p) {
                const int q = p * 32 + ln2;
                const int row = q < 90 ? q : 89;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb2 * 4;
                    const int off = row * RB + (((ch >> 3) ^ (row & 15)) << 4) + (ch & 7) * 2;
                    Quad<_Float16> hq;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                        hq.e[i] = (_Float16)r;
                        acc[p][g * 4 + i] = r;
                        lo[p][g * 4 + i] = r - (float)hq.e[i];
                    }
                    if (q < 90) *reinterpret_cast<Quad<_Float16>*>(Y + off) = hq;
                }
            }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {                   // pair (0, 1), then tile 2 with itself
                const int pa = pp == 0 ? 0 : 2, pb = pp == 0 ? 1 : 2;
                f32x16 av, bv, al, bl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[pa][r]), __float_as_uint(acc[pb][r]), false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[pa][r]), __float_as_uint(lo[pb][r]), false, false);
                    av[r] = __uint_as_float(sv[0]); bv[r] = __uint_as_float(sv[1]);
                    al[r] = __uint_as_float(sl[0]); bl[r] = __uint_as_float(sl[1]);
                }
                const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
                const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
                const int q = pp == 0 ? kb2 * 32 + ln2 : 64 + ln2;
                if (pp == 0 || (kb2 == 0 && q < 90)) {
                    unsigned char* d0 = Y + PART + c6_lds_off(q, c6_chunk(0, wave));
                    unsigned char* t0 = Y + PART + c6_lds_off(q, c6_chunk(0, wave) + 1) + c6_tail_half(q);
                    unsigned char* d1 = Y + PART + c6_lds_off(q, c6_chunk(1, wave));
                    unsigned char* t1 = Y + PART + c6_lds_off(q, c6_chunk(1, wave) + 1) + c6_tail_half(q);
                    *reinterpret_cast<uint4*>(d0) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                    *reinterpret_cast<uint2*>(t0) = make_uint2(pl[4], pl[5]);
                    *reinterpret_cast<uint4*>(d1) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                    *reinterpret_cast<uint2*>(t1) = make_uint2(pv[4], pv[5]);
                }
            }
            // the accumulators restart at b2 + skip: this lane's channels of X (f16 quads + its elements of the lo piece)
#pragma unroll
            for (int p = 0; p < NT; ++p) {
                const int q = p * 32 + ln2;
                const int row = q < 90 ? q : 89;
                f32x32 xl;
                if (!FIRST) {
                    const uint4 hd4 = *reinterpret_cast<const uint4*>(X + PART + c6_lds_off(row, c6_chunk(0, wave)));
                    const uint2 tl2 = *reinterpret_cast<const uint2*>(X + PART + c6_lds_off(row, c6_chunk(0, wave) + 1) + c6_tail_half(row));
                    // the upper lane half wants the odd elements: it shifts the piece down by one element (6 bits), so that
                    // every lane reads element 2 r for its register r  (a select between xl[2 r] and xl[2 r + 1] is
                    // turned into a variable vector index by the compiler: 31 v_cndmask per element)
                    const uint32_t wv[7] = {hd4.x, hd4.y, hd4.z, hd4.w, tl2.x, tl2.y, 0u};
                    const uint32_t sh6 = (uint32_t)kb2 * 6u;
                    u32x6 pc;
#pragma unroll
                    for (int w = 0; w < 6; ++w) pc[w] = __builtin_amdgcn_alignbit(wv[w + 1], wv[w], sh6);
                    xl = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(pc, s_skip);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb2 * 4;
                    const int off = row * RB + (((ch >> 3) ^ (row & 15)) << 4) + (ch & 7) * 2;
                    const float4 bv4 = *reinterpret_cast<const float4*>(bias2 + ch);
                    float vv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
                    const Quad<_Float16> xq = *reinterpret_cast<const Quad<_Float16>*>(X + off);
                    if (FIRST) {
                        const int off_lo = PART + row * RB + (ch & 15) + (((ch >> 4) ^ (row & 15)) << 4);
                        cf8::add_pair4(vv, xq, *reinterpret_cast<const uint32_t*>(X + off_lo));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int r = g * 4 + i;
                            vv[i] += (float)xq.e[i] + xl[2 * r];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[p][g * 4 + i] = vv[i];
                }
            }
        } else {
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            const int row = q < 90 ? q : 89;                   // (padding lanes compute on row 89 and store nothing)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wave * 32 + g * 8 + kb2 * 4;
                const int off = row * RB + (((ch >> 3) ^ (row & 15)) << 4) + (ch & 7) * 2;
                const int off_lo = PART + row * RB + (ch & 15) + (((ch >> 4) ^ (row & 15)) << 4);
                const int off_hi = PART + row * RB + (ch & 15) + (((8 + (ch >> 4)) ^ (row & 15)) << 4);
                float r[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                const cf8::Split4 o = cf8::split4(r);
                if (q < 90) {
                    *reinterpret_cast<Quad<_Float16>*>(Y + off) = o.hi;
                    *reinterpret_cast<uint32_t*>(Y + off_lo) = o.l8;
                    *reinterpret_cast<uint32_t*>(Y + off_hi) = o.h8;
                }
                const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
                float vv[4] = {bv.x, bv.y, bv.z, bv.w};
                cf8::add_pair4(vv, *reinterpret_cast<const Quad<_Float16>*>(X + off), *reinterpret_cast<const uint32_t*>(X + off_lo));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[p][g * 4 + i] = vv[i];
            }
        }
        }
        RB_STAMP(3);
        __syncthreads();                                       // B_k: Y complete, X and the previous staging contents free
        RB_STAMP(4);
        __builtin_amdgcn_s_setprio(3);
        c8k::kloop<NT, c8k::NoShadow, 0, false, 128, C6 ? 1 : 0>(Y, c8k::Image{0, ZROW, PART}, flt2, lane, acc, 127 + k_y - cf8::X_LO_SHIFT, 127 + k_y);
        __builtin_amdgcn_s_setprio(0);
        RB_STAMP(5);
#pragma unroll
        for (int p = 0; p < NT; ++p) prev[p] = acc[p];
        have_prev = true;
        if (!has_next) break;
        t += stride;
    }
    // the last board's second epilogue, not overlapped (once the copy waves have let go of the staging image)
    __syncthreads();                                           // E0
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) shd.unit(prev[p], p * 32 + ln, g);
    __syncthreads();                                           // E1
}

// (kernel 2d, the chains of residual blocks in one launch -- k_tower, k_tower_pairs -- live in csrc/xq_tower.hip)

// ---- kernel 2b: the residual block, software-pipelined over boards (128 filters, split operands) ------------------
// Same arithmetic as k_resblock (bit-identical results), different schedule: the second epilogue of board t-1
// (+ bias2 + skip, ReLU, re-split) runs in the SHADOW of board t's first K loop instead of between the K loops, and
// writes its result IN PLACE over the skip operand, in the operand layout -- so there is no fp32 staging buffer, no
// conversion in the copy waves, no hand-over of X between the roles, and two barriers per board instead of three.
//   LDS: three images (X even, X odd, Y) of 90 rows x 256 B per part, 16 zero rows shared by all of them, the two bias
//   vectors: 2 x (270 + 2 + 16) rows x 256 B + 1 KB = 148.5 KB.  Absolute row r of part q is byte (q * PSTR + r * 256).
//   matrix waves, board k (X = image k & 1):  A | K1(k) with epi2(k-1) in its shadow | epi1(k) -> Y | B | K2(k) | ...
//   copy waves:                               A | prefetch board k+1 (registers)     |             | B | drain out(k-1)
//                                                 from image (k-1) & 1 to HBM and refill the same chunks with board k+1
// The shadow work is cut into 12 units (pixel tile p x channel group g); K loop 1 is a rolled loop over 3 x (3 taps =
// 216 MFMA slots) and each pass retires the 4 units of one pixel tile, whose accumulators are then rotated out, so
// every register index in the loop body is static.
// FIRST: the block is the first of the tower and its input is the 5 x 5 input convolution of the feature planes
// (Conv2D(F, 5) -> BatchNorm -> ReLU, agent/model.py:36-39), computed HERE by the copy waves instead of by a kernel of
// its own.  The planes are one-hot -- a position has at most 32 pieces, 64 plane bits with the history planes -- so the
// layer is a gather: output pixel p, channel o = bias[o] + sum over the occupied squares q of p's 5 x 5 window of
// w[o][plane at q][tap].  A copy thread owns the 8 channels of its 16-byte chunk for 6 pixels (the chunks it writes into
// the X image anyway): 48 fp32 accumulators, ~55 weight loads of 32 bytes from an L2-resident table
// (table[plane][tap][128], 179 KB) per board, exact fp32 sums in a fixed order (bias, taps 0..24, planes ascending) --
// no MFMA, no second kernel, no 46 KB per board written and read back.  The work is split over the two windows a copy
// wave has per board (under K loop 1 and under K loop 2), since all waves meet at the barrier between them.

template <typename E, bool FIRST = false>
__global__ __launch_bounds__(512, 1) void k_resblock_pipe(
    const E* __restrict__ xh, const E* __restrict__ xl, const E* __restrict__ w1p, const float* __restrict__ b1,
    const E* __restrict__ w2p, const float* __restrict__ b2, E* __restrict__ yh, E* __restrict__ yl, int n_boards,
    const int32_t* __restrict__ n_dev, FirstArgs fa)
{
    using namespace pipe;
    constexpr int C = 128, CHUNKS = 90 * 16, CTHR = 256, LITER = (CHUNKS + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = gridDim.x;
    int t = blockIdx.x;
    if (t >= n_boards) return;
    // chunk i of a board: row i / 16, chunk i % 16 -> LDS byte (image row base + row) * 256 + ((ch ^ (row & 15)) << 4)
    auto chunk_off = [&](int img, int i) {
        const int row = i >> 4, ch = i & 15;
        return (img * IMG_ROWS + row) * RB + ((ch ^ (row & 15)) << 4);
    };

    if (wave >= 4) {                                   // ---- copy waves ----
        const int ctid = tid - 256;
        uint4 v[2][LITER];
        auto fetch = [&](int board) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const uint4* src = reinterpret_cast<const uint4*>((part ? xl : xh) + (size_t)board * 90 * C);
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    v[part][it] = make_uint4(0, 0, 0, 0);
                    if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS) v[part][it] = src[i];
                }
            }
        };
        auto put = [&](int img) {
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS)
                        *reinterpret_cast<uint4*>(lds + part * PSTR + chunk_off(img, i)) = v[part][it];
                }
        };
        // result of `board` sits in image img (operand layout): to HBM; then the same chunks take the prefetched board
        auto drain = [&](int img, int board, bool refill) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                uint4* dst = reinterpret_cast<uint4*>((part ? yl : yh) + (size_t)board * 90 * C);
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    if (!((it + 1) * CTHR <= CHUNKS || i < CHUNKS)) continue;
                    unsigned char* a = lds + part * PSTR + chunk_off(img, i);
                    const uint4 o = *reinterpret_cast<const uint4*>(a);
                    if (refill) *reinterpret_cast<uint4*>(a) = v[part][it];
                    dst[i] = o;
                }
            }
        };
        // ---- FIRST: the input layer of board `board` into v[][] (the chunk layout put() / drain() write), in two halves ----
        float acc[FIRST ? LITER : 1][8];
        uint32_t occ[FIRST ? LITER : 1], cur_m[FIRST ? LITER : 1];
        int cur_tap[FIRST ? LITER : 1];
        uint32_t* mk = reinterpret_cast<uint32_t*>(lds + MASK_OFF) + (wave - 4) * 96;     // this wave's own mask board
        const int c8 = ctid & 15, prow = ctid >> 4;            // chunk = channels 8 c8 .. 8 c8 + 7 of pixels prow + 16 it
        constexpr int PW = 12;                                 // plane words per lane: 32 planes x 90 bytes / 4 / 64 lanes
        uint32_t pw[FIRST ? PW : 1];
        auto planes_prefetch = [&](int board) {                // HBM -> registers, consumed by the next first_begin
            const int per_board = fa.in_planes * 90;
            const size_t slot = (size_t)(fa.rows ? fa.rows[board] : board);
            if (fa.masks) {                                    // the board's occupancy board, as the search kernel wrote it
                pw[0] = fa.masks[slot * 96 + lane];
                pw[1] = lane < 32 ? fa.masks[slot * 96 + 64 + lane] : 0u;
                return;
            }
            const uint32_t* src = reinterpret_cast<const uint32_t*>(fa.planes + slot * per_board);
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                const int w = lane + 64 * j;
                pw[j] = w < per_board / 4 ? src[w] : 0u;
            }
        };
        auto first_begin = [&]() {
            // the occupied planes of every square as a bit mask: handed in ready-made (fa.masks, cz_search_leaf_masks), or
            // built by each copy wave for itself (no barrier with the other copy waves: they share nothing)
            if (fa.masks) {
                mk[lane] = pw[0];
                if (lane < 32) mk[64 + lane] = pw[1];
            } else {
                mk[lane] = 0u;
                if (lane < 32) mk[64 + lane] = 0u;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // (the board's plane words are already in registers: planes_prefetch ran a window earlier)
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    const int w = lane + 64 * j;
                    const uint32_t word = pw[j];
                    if (word == 0u) continue;
                    int c = (w * 4) / 90, pix = w * 4 - c * 90;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        if ((word >> (8 * k4)) & 0xFFu) atomicOr(&mk[pix], 1u << c);
                        if (++pix == 90) { pix = 0; ++c; }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // the occupied squares as one 90-bit set (wave-uniform): a pixel's occupied taps are five 5-bit windows of it
            const uint64_t occ_lo = __ballot(mk[lane] != 0u);
            const uint64_t occ_hi = __ballot(lane < 26 && mk[64 + (lane & 31)] != 0u);
            const float4 b0 = *reinterpret_cast<const float4*>(fa.in_bias + c8 * 8);
            const float4 b1v = *reinterpret_cast<const float4*>(fa.in_bias + c8 * 8 + 4);
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                acc[it][0] = b0.x; acc[it][1] = b0.y; acc[it][2] = b0.z; acc[it][3] = b0.w;
                acc[it][4] = b1v.x; acc[it][5] = b1v.y; acc[it][6] = b1v.z; acc[it][7] = b1v.w;
                // the taps of pixel `it` whose square is occupied, as a 25-bit set
                const int p = prow + 16 * it;
                const int py = p / 9, px = p - py * 9;
                uint32_t o = 0u;
                if (p < 90) {
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) {
                        const int r = py + ky - 2;
                        if ((unsigned)r < 10u) {
                            // the 9 squares of board row r, then columns px - 2 .. px + 2 (columns off the board shift in zeros)
                            const int sh = r * 9;
                            const uint32_t rowbits = (uint32_t)((sh < 64 ? (occ_lo >> sh) | (sh > 55 ? occ_hi << (64 - sh) : 0ull)
                                                                         : occ_hi >> (sh - 64)) & 0x1FFull);
                            o |= (((rowbits << 2) >> px) & 31u) << (5 * ky);
                        }
                    }
                }
                occ[it] = o;
                cur_m[it] = 0u;
                cur_tap[it] = 0;
            }
        };
        // One round = the next (tap, plane) term of each of the thread's six pixels: up to twelve 16-byte loads in flight,
        // then the adds.  A pixel's terms are taken taps ascending, planes ascending -- the summation order does not
        // depend on how the rounds fall.  Iterating over the terms a pixel HAS (9 on average, 25 at most) instead of over
        // the 25 taps halves the number of L2 round trips, which is all this loop costs.
        auto first_rounds = [&](int max_rounds) {
#pragma unroll 1
            for (int round = 0; round < max_rounds; ++round) {
                uint32_t any = 0u;
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    if (cur_m[it] == 0u && occ[it] != 0u) {
                        const int tap = __builtin_ctz(occ[it]);
                        occ[it] &= occ[it] - 1u;
                        const int p = prow + 16 * it;
                        cur_tap[it] = tap;
                        cur_m[it] = mk[p + (tap / 5 - 2) * 9 + (tap % 5 - 2)];
                    }
                    any |= cur_m[it];
                }
                if (!__ballot(any != 0u)) break;
                float4 wa[LITER], wb[LITER];
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    wa[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    wb[it] = wa[it];
                    if (cur_m[it]) {
                        const int c = __builtin_ctz(cur_m[it]);
                        const float4* tp = reinterpret_cast<const float4*>(fa.table + ((size_t)(c * 25 + cur_tap[it]) * 128 + c8 * 8));
                        wa[it] = tp[0];
                        wb[it] = tp[1];
                    }
                }
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    if (cur_m[it]) {
                        acc[it][0] += wa[it].x; acc[it][1] += wa[it].y; acc[it][2] += wa[it].z; acc[it][3] += wa[it].w;
                        acc[it][4] += wb[it].x; acc[it][5] += wb[it].y; acc[it][6] += wb[it].z; acc[it][7] += wb[it].w;
                        cur_m[it] &= cur_m[it] - 1u;
                    }
                }
            }
        };
        auto first_end = [&]() {                              // ReLU, split, pack: v[part][it] = chunk (pixel prow + 16 it, c8)
            struct alignas(16) E8 { E e[8]; };
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                E8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float r = acc[it][j] > 0.0f ? acc[it][j] : 0.0f;
                    hi.e[j] = (E)r;
                    lo.e[j] = (E)(r - (float)hi.e[j]);
                }
                v[0][it] = __builtin_bit_cast(uint4, hi);
                v[1][it] = __builtin_bit_cast(uint4, lo);
            }
        };
        if (FIRST) {
            planes_prefetch(t);
            first_begin();
            if (t + stride < n_boards) planes_prefetch(t + stride);
            first_rounds(25 * 32);       // (to the end: a pixel has at most 25 taps x in_planes terms)
            first_end();
        } else {
            fetch(t);
        }
        put(0);
        for (int i = ctid; i < 16 * 16; i += CTHR) {          // the shared zero rows, both parts
            *reinterpret_cast<uint4*>(lds + ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + PSTR + ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
        }
        if (ctid < 128) {
            reinterpret_cast<float*>(lds + BIAS_OFF)[ctid] = b1[ctid];
            reinterpret_cast<float*>(lds + BIAS_OFF)[128 + ctid] = b2[ctid];
        }
        int k = 0, t_prev = -1;
        for (;;) {
            __syncthreads();                                   // A_k
            const int tn = t + stride;
            const bool has_next = tn < n_boards;
            if (FIRST) {
                if (has_next) {                                // first half of the next board's input layer, under K loop 1
                    first_begin();                             // (board tn: its planes were fetched a window ago)
                    if (tn + stride < n_boards) planes_prefetch(tn + stride);
                    first_rounds(fa.w1_rounds);
                }
            } else if (has_next) {
                fetch(tn);
            }
            __syncthreads();                                   // B_k: out(k-1) complete in image (k-1) & 1
            if (FIRST) {
                if (t_prev >= 0) drain((k - 1) & 1, t_prev, false);
                if (has_next) {                                // second half, under K loop 2; then into the free image
                    first_rounds(25 * 32);       // (to the end: a pixel has at most 25 taps x in_planes terms)
                    first_end();
                    put((k - 1) & 1);
                }
            } else {
                if (t_prev >= 0) drain((k - 1) & 1, t_prev, has_next);
                else if (has_next) put(1);
            }
            t_prev = t;
            if (!has_next) break;
            t = tn;
            ++k;
        }
        __syncthreads();                                       // E1: K2 of the last board done
        __syncthreads();                                       // E2: its epilogue 2 written in place
        drain(k & 1, t_prev, false);
        return;
    }

    // ---- matrix waves ----
    const uint4* wq1 = reinterpret_cast<const uint4*>(w1p) + wave * 64 + lane;
    const uint4* wq2 = reinterpret_cast<const uint4*>(w2p) + wave * 64 + lane;
    const int kb = lane >> 5, ln = lane & 31;
    f32x16 acc[NT], prev[NT];
    PipeShadow<E> shd;
    shd.lds = lds; shd.wave = wave; shd.kb = kb; shd.ln = ln;
    int k = 0;
    for (;;) {
        __syncthreads();                                       // A_k: X(k) in image k & 1
        const bool has_next = t + stride < n_boards;
        shd.prev_row_base = ((k - 1) & 1) * IMG_ROWS;
        __builtin_amdgcn_s_setprio(3);
        if (k > 0) pipe_kloop<E, true>(lds, (k & 1) * IMG_ROWS, wq1, lane, acc, prev, shd);
        else pipe_kloop<E, false>(lds, (k & 1) * IMG_ROWS, wq1, lane, acc, prev, shd);
        __builtin_amdgcn_s_setprio(0);
        int ln2 = ln, kb2 = kb;
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 1: relu(acc + b1) -> (hi, lo) -> Y (image 2)
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            if (q < 90) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb2 * 4;
                    const float4 bv = *reinterpret_cast<const float4*>(lds + BIAS_OFF + ch * 4);
                    const float vv[4] = {acc[p][g * 4 + 0] + bv.x, acc[p][g * 4 + 1] + bv.y, acc[p][g * 4 + 2] + bv.z,
                                         acc[p][g * 4 + 3] + bv.w};
                    Quad<E> hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                        hi.e[i] = (E)r;
                        lo.e[i] = (E)(r - (float)hi.e[i]);
                    }
                    const int off = (2 * IMG_ROWS + q) * RB + (((ch >> 3) ^ (q & 15)) << 4) + (ch & 7) * 2;
                    *reinterpret_cast<Quad<E>*>(lds + off) = hi;
                    *reinterpret_cast<Quad<E>*>(lds + PSTR + off) = lo;
                }
            }
        }
        __syncthreads();                                       // B_k: Y complete (and out(k-1), written during K1)
        __builtin_amdgcn_s_setprio(3);
        pipe_kloop<E, false>(lds, 2 * IMG_ROWS, wq2, lane, acc, prev, shd);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int p = 0; p < NT; ++p) prev[p] = acc[p];
        if (!has_next) break;
        t += stride;
        ++k;
    }
    __syncthreads();                                           // E1
    shd.prev_row_base = (k & 1) * IMG_ROWS;                    // epilogue 2 of the last board, not overlapped
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            shd.whole(prev[p], p, g);
        }
    __syncthreads();                                           // E2
}

// ---- kernel 2c: the residual block with split operands where three LDS images do not fit (192 filters) --------------
// The reference's deployed topology is 10 x 192 (configs/distribute.py:84-87, data/model/model_192x10_config.json).
// With split operands a 192-filter image is 90 rows x 384 B x 2 parts = 69 KB: X + Y + a staging image (k_resblock) or
// X even + X odd + Y (k_resblock_pipe) need 207 KB, the CU has 160.  This kernel keeps TWO images -- X and Y, 16 zero
// rows shared by both, the bias vectors: 152 KB -- and gives up the overlap that the third image buys:
//   matrix waves (6, 32 output channels each):  A | K1 on X | epi1 -> Y | B | K2 on Y | epi2: + b2 + skip, ReLU, re-split,
//                                               written IN PLACE over the skip operand in X | C |
//   copy waves (2):                             A | prefetch the next board into registers       | B | C | drain X to
//                                               HBM (16 B per lane, whole rows) and refill it with the prefetched board
// so HBM sees one read of x and one write of y per block (two launches of k_conv3x3 read x twice -- operand and skip --
// and round-trip the intermediate activation, with 8-byte scattered stores), and the matrix waves wait only for the
// drain + refill of one image (LDS <-> registers; the HBM latency of the prefetch is under the K loops).
// Same K loop (conv_kloop) and the same epilogue arithmetic in the same order as k_conv3x3: bit-identical to two
// cz_conv3x3 launches.  The last block of a tower writes fp32 (y_f32) straight from the accumulators' epilogue instead.
namespace ip {
constexpr int ROW_Y = 90, ROW_Z = 192, ROWS = ROW_Z + 16, COPY_THREADS = 128;       // (ROW_Z: a multiple of 16)
constexpr int MAX_BLOCKS = 12;
struct Chain {                      // k_resblock_ip_c8: the consecutive blocks a launch takes every board through (1 .. 12)
    const void* w1[MAX_BLOCKS];
    const void* w2[MAX_BLOCKS];
    const float* b1[MAX_BLOCKS];
    const float* b2[MAX_BLOCKS];
    int n;
};
}

template <typename E, int C>
__global__ __launch_bounds__((C / 32) * 64 + ip::COPY_THREADS, 1) void k_resblock_ip(
    const E* __restrict__ xh, const E* __restrict__ xl, const E* __restrict__ w1p, const float* __restrict__ b1,
    const E* __restrict__ w2p, const float* __restrict__ b2, E* __restrict__ yh, E* __restrict__ yl,
    float* __restrict__ yf, int n_boards, const int32_t* __restrict__ n_dev)
{
    typedef Geom<C, 1, 2> G;
    constexpr int RB = G::RB, CPR = G::CPR, CT = G::CT, NT = 3;
    constexpr int PSTR = ip::ROWS * RB;                         // bytes per operand part
    constexpr int BIAS_OFF = 2 * PSTR;
    constexpr int CTHR = ip::COPY_THREADS, CHUNKS = 90 * CPR, LITER = (CHUNKS + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[BIAS_OFF + 2 * C * 4];
    static_assert(sizeof(lds) <= 160 * 1024, "two images + zero rows must fit the CU's LDS");
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = gridDim.x;
    int t = blockIdx.x;
    if (t >= n_boards) return;
    // chunk i of a board (row i / CPR, chunk i % CPR) in the X image
    auto chunk_off = [&](int i) {
        const int row = i / CPR, ch = i - row * CPR;
        return row * RB + ((ch ^ (row & G::SWZ)) << 4);
    };

    if (wave >= CT) {                                           // ---- copy waves ----
        const int ctid = tid - CT * 64;
        uint4 v[2][LITER];
        auto fetch = [&](int board) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const uint4* src = reinterpret_cast<const uint4*>((part ? xl : xh) + (size_t)board * 90 * C);
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    v[part][it] = make_uint4(0, 0, 0, 0);
                    if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS) v[part][it] = src[i];
                }
            }
        };
        auto put = [&]() {
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS)
                        *reinterpret_cast<uint4*>(lds + part * PSTR + chunk_off(i)) = v[part][it];
                }
        };
        // the block's result sits in X (operand layout): to HBM; the same chunks then take the prefetched board
        auto drain = [&](int board, bool refill) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                uint4* dst = reinterpret_cast<uint4*>((part ? yl : yh) + (size_t)board * 90 * C);
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    if (!((it + 1) * CTHR <= CHUNKS || i < CHUNKS)) continue;
                    unsigned char* a = lds + part * PSTR + chunk_off(i);
                    if (!yf) dst[i] = *reinterpret_cast<const uint4*>(a);
                    if (refill) *reinterpret_cast<uint4*>(a) = v[part][it];
                }
            }
        };
        fetch(t);
        put();
        for (int i = ctid; i < 16 * CPR; i += CTHR) {           // the shared zero rows, both parts
            *reinterpret_cast<uint4*>(lds + ip::ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + PSTR + ip::ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
        }
        for (int i = ctid; i < C; i += CTHR) {
            reinterpret_cast<float*>(lds + BIAS_OFF)[i] = b1[i];
            reinterpret_cast<float*>(lds + BIAS_OFF)[C + i] = b2[i];
        }
        for (;;) {
            __syncthreads();                                    // A: X holds board t
            const int tn = t + stride;
            const bool has_next = tn < n_boards;
            if (has_next) fetch(tn);
            __syncthreads();                                    // B: Y complete
            __syncthreads();                                    // C: the result is in X
            drain(t, has_next);
            if (!has_next) break;
            t = tn;
        }
        return;
    }

    // ---- matrix waves ----
    const uint4* wq1 = reinterpret_cast<const uint4*>(w1p) + wave * 64 + lane;
    const uint4* wq2 = reinterpret_cast<const uint4*>(w2p) + wave * 64 + lane;
    const int kb = lane >> 5, ln = lane & 31;
    const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF);
    const float* bias2 = bias1 + C;
    for (;;) {
        __syncthreads();                                        // A
        const bool has_next = t + stride < n_boards;
        f32x16 acc[NT];
        __builtin_amdgcn_s_setprio(3);
        conv_kloop<E, C, 1, 2>(lds, wq1, lane, acc, 0, ip::ROW_Z, PSTR);
        __builtin_amdgcn_s_setprio(0);
        int ln2 = ln, kb2 = kb;
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 1: relu(acc + b1) -> (hi, lo) -> Y image (operand layout of the second convolution)
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            if (q < 90) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb2 * 4;
                    const float4 bv = *reinterpret_cast<const float4*>(bias1 + ch);
                    const float vv[4] = {acc[p][g * 4 + 0] + bv.x, acc[p][g * 4 + 1] + bv.y, acc[p][g * 4 + 2] + bv.z,
                                         acc[p][g * 4 + 3] + bv.w};
                    Quad<E> hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                        hi.e[i] = (E)r;
                        lo.e[i] = (E)(r - (float)hi.e[i]);
                    }
                    const int off = (ip::ROW_Y + q) * RB + (((ch >> 3) ^ (q & G::SWZ)) << 4) + (ch & 7) * 2;
                    *reinterpret_cast<Quad<E>*>(lds + off) = hi;
                    *reinterpret_cast<Quad<E>*>(lds + PSTR + off) = lo;
                }
            }
        }
        __syncthreads();                                        // B: Y complete
        __builtin_amdgcn_s_setprio(3);
        conv_kloop<E, C, 1, 2>(lds, wq2, lane, acc, ip::ROW_Y, ip::ROW_Z, PSTR);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 2: relu(acc + b2 + x) -> (hi, lo) in place over the skip operand (each lane reads and writes only its own
        // 8 bytes of each part), or fp32 straight to HBM for the last block of a tower
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            if (q < 90) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb2 * 4;
                    const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
                    const int off = q * RB + (((ch >> 3) ^ (q & G::SWZ)) << 4) + (ch & 7) * 2;
                    const Quad<E> sh = *reinterpret_cast<const Quad<E>*>(lds + off);
                    const Quad<E> sl = *reinterpret_cast<const Quad<E>*>(lds + PSTR + off);
                    float vv[4] = {acc[p][g * 4 + 0] + bv.x, acc[p][g * 4 + 1] + bv.y, acc[p][g * 4 + 2] + bv.z,
                                   acc[p][g * 4 + 3] + bv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) vv[i] += (float)sh.e[i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) vv[i] += (float)sl.e[i];
#pragma unroll
                    for (int i = 0; i < 4; ++i) vv[i] = vv[i] > 0.0f ? vv[i] : 0.0f;
                    if (yf) {
                        *reinterpret_cast<float4*>(yf + ((size_t)t * 90 + q) * C + ch) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    } else {
                        Quad<E> hi, lo;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            hi.e[i] = (E)vv[i];
                            lo.e[i] = (E)(vv[i] - (float)hi.e[i]);
                        }
                        *reinterpret_cast<Quad<E>*>(lds + off) = hi;
                        *reinterpret_cast<Quad<E>*>(lds + PSTR + off) = lo;
                    }
                }
            }
        }
        __syncthreads();                                        // C: the result is in X
        if (!has_next) break;
        t += stride;
    }
}

// ---- kernel 2c / c8: the two-image residual block on the c8 arithmetic (192 filters; round 4) --------------------------------
// k_resblock_ip's roles, images and barriers (X | Y | 16 shared zero rows; six matrix waves + two copy waves; the second
// epilogue in place over X; the copy waves drain X to HBM and refill it at barrier C) with c8k::kloop<..., 192> -- 384-byte
// pixel rows [f16 x 192] / [lo8 x 192 | e4m3(x) x 192], swizzled inside aligned groups of eight chunks, three 64-channel
// blocks per tap -- and k_resblock_c8's arithmetic: accumulators start at bias (K loop 1) / bias + skip (K loop 2, read from
// X by the owning lane), the epilogues only apply ReLU and split.  2.0 instead of 3.0 MFMA-equivalents per product for the
// reference's deployed 10 x 192 topology (configs/distribute.py:84-87).  Bit-identical to two cz_conv3x3_c8 launches.
// XF / YF (round 6): the formats of the image the first convolution reads and of the intermediate image -- 0 = c8 (e4m3
// corrections), 1 = c6 (bf6 pieces with an exponent: cz_conv3x3_c6_pack_weights filters; 25 % less matrix time per product).
// <0, 0>: the c8 block; <1, 1>: a c6 block; <0, 1>: the tower's first c6 block, whose input is the input layer's c8 image.
// A c6 block's result is a c6 image with its second filter's output exponent -- or a c8 image where that filter carries
// y_exp = 127 (the hand-over of a c6>N tower), or fp32 (y_f32).  The c6 pieces are formed exactly as in k_resblock_c8<.., C6>
// (tiles trade lane halves with v_permlane32_swap, v_cvt_scalef32_2xpk16_bf6_f32); piece (kind, 32-channel block w) of a pixel
// row has its 16-byte head in chunk kind * CPR / 2 + 4 (w >> 1) + 2 (w & 1), its 8-byte tail at the start of the next chunk.
// CHAIN (round 6, cz_resblock_chain): ch.n consecutive blocks of ONE format per launch.  Two images leave no room for a second
// board, so a workgroup takes ONE board through all blocks: the in-place second epilogue already leaves block b's result in X
// as block b + 1's input, and the drain to HBM + refill (69 KB through two copy waves, all matrix waves waiting) happens once
// per chain instead of once per block.  Biases double-buffered in LDS by the running block count; filters and image exponents
// switch per block; y_f32 applies to the chain's last block.  Same arithmetic as ch.n one-block launches: bit-identical.
template <int C, int XF = 0, int YF = 0>
__global__ __launch_bounds__((C / 32) * 64 + ip::COPY_THREADS, 1) void k_resblock_ip_c8(
    const _Float16* __restrict__ xh, const unsigned char* __restrict__ xc, ip::Chain ch, _Float16* __restrict__ yh,
    unsigned char* __restrict__ yc, float* __restrict__ yf_last, int n_boards, const int32_t* __restrict__ n_dev)
{
    typedef Geom<C, 1, 2> G;
    constexpr int RB = G::RB, CPR = G::CPR, CT = G::CT, NT = 3;
    constexpr int PSTR = ip::ROWS * RB;                         // bytes per operand part
    constexpr int BIAS_OFF = 2 * PSTR;
    constexpr int CTHR = ip::COPY_THREADS, CHUNKS = 90 * CPR, LITER = (CHUNKS + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[BIAS_OFF + 2 * 2 * C * 4];       // bias[2 buffers][2 convolutions][C]
    const int NB = ch.n;
    static_assert(sizeof(lds) <= 160 * 1024, "two images + zero rows must fit the CU's LDS");
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = gridDim.x;
    int t = blockIdx.x;
    if (t >= n_boards) return;
    auto chunk_off = [&](int i) {                               // chunk i of a board (row i / CPR, chunk i % CPR) in the X image
        const int row = i / CPR, ch = i - row * CPR;
        return row * RB + ((ch & ~G::SWZ) << 4) + (((ch ^ row) & G::SWZ) << 4);
    };

    if (wave >= CT) {                                           // ---- copy waves ----
        const int ctid = tid - CT * 64;
        uint4 v[2][LITER];
        auto fetch = [&](int board) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const uint4* src = part ? reinterpret_cast<const uint4*>(xc + (size_t)board * 90 * 2 * C)
                                        : reinterpret_cast<const uint4*>(xh + (size_t)board * 90 * C);
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    v[part][it] = make_uint4(0, 0, 0, 0);
                    if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS) v[part][it] = src[i];
                }
            }
        };
        auto put = [&]() {
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS)
                        *reinterpret_cast<uint4*>(lds + part * PSTR + chunk_off(i)) = v[part][it];
                }
        };
        auto drain = [&](int board, bool refill) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                uint4* dst = part ? reinterpret_cast<uint4*>(yc + (size_t)board * 90 * 2 * C)
                                  : reinterpret_cast<uint4*>(yh + (size_t)board * 90 * C);
#pragma unroll
                for (int it = 0; it < LITER; ++it) {
                    const int i = it * CTHR + ctid;
                    if (!((it + 1) * CTHR <= CHUNKS || i < CHUNKS)) continue;
                    unsigned char* a = lds + part * PSTR + chunk_off(i);
                    if (!yf_last) dst[i] = *reinterpret_cast<const uint4*>(a);
                    if (refill) *reinterpret_cast<uint4*>(a) = v[part][it];
                }
            }
        };
        fetch(t);
        put();
        for (int i = ctid; i < 16 * CPR; i += CTHR) {           // the shared zero rows, both parts
            *reinterpret_cast<uint4*>(lds + ip::ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + PSTR + ip::ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
        }
        // biases of running block g (block g % NB) into buffer g & 1
        auto write_bias = [&](int g) {
            const int blk = g % NB;
            float* dst = reinterpret_cast<float*>(lds + BIAS_OFF) + (g & 1) * 2 * C;
            for (int i = ctid; i < C; i += CTHR) {
                dst[i] = ch.b1[blk][i];
                dst[C + i] = ch.b2[blk][i];
            }
        };
        write_bias(0);
        write_bias(1);                                          // (one block per launch: both buffers hold its biases)
        int g = 0;
        for (;;) {
            __syncthreads();                                    // A: X holds board t
            const int tn = t + stride;
            const bool has_next = tn < n_boards;
            if (has_next) fetch(tn);
            for (int b = 0; b < NB; ++b, ++g) {
                __syncthreads();                                // B: Y complete; block g's biases are consumed
                if (NB > 1) write_bias(g + 2);                  // (one block per launch: its biases never change)
                __syncthreads();                                // C: the block's result is in X
            }
            drain(t, has_next);
            if (!has_next) break;
            t = tn;
        }
        return;
    }

    // ---- matrix waves ----
    const int kb = lane >> 5, ln = lane & 31;
    // byte offset (inside a part) of 16-byte chunk `chunk` of pixel row `row_abs`, swizzle key = the image-relative row
    auto choff = [&](int row_abs, int key, int chunk) {
        return row_abs * RB + ((chunk & ~G::SWZ) << 4) + (((chunk ^ key) & G::SWZ) << 4);
    };
    // relu'd fp32 accumulators of the three tiles -> the c6 operand triple of image `row0` (its first absolute row) with
    // exponent k: f16 quads per lane, the two bf6 pieces of a pixel's 32 channels after the lane halves traded tiles.
    // acc keeps relu(acc).
    auto write_c6 = [&](f32x16* acc, int row0, int k, int ln2, int kb2) __attribute__((always_inline)) {
        using rb8::u32x6;
        const float s_hi = __builtin_ldexpf(1.0f, k), s_lo = __builtin_ldexpf(1.0f, k - cf8::X_LO_SHIFT);
        f32x16 lo[NT];
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            const int row = q < 90 ? q : 89;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wave * 32 + g * 8 + kb2 * 4;
                Quad<_Float16> hq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                    hq.e[i] = (_Float16)r;
                    acc[p][g * 4 + i] = r;
                    lo[p][g * 4 + i] = r - (float)hq.e[i];
                }
                if (q < 90) *reinterpret_cast<Quad<_Float16>*>(lds + choff(row0 + row, row, ch >> 3) + (ch & 7) * 2) = hq;
            }
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {                       // pair (0, 1), then tile 2 with itself
            const int pa = pp == 0 ? 0 : 2, pb = pp == 0 ? 1 : 2;
            f32x16 av, bv, al, bl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[pa][r]), __float_as_uint(acc[pb][r]), false, false);
                const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[pa][r]), __float_as_uint(lo[pb][r]), false, false);
                av[r] = __uint_as_float(sv[0]); bv[r] = __uint_as_float(sv[1]);
                al[r] = __uint_as_float(sl[0]); bl[r] = __uint_as_float(sl[1]);
            }
            const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
            const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
            const int q = pp == 0 ? kb2 * 32 + ln2 : 64 + ln2;
            if (pp == 0 || (kb2 == 0 && q < 90)) {
                const int c0 = 4 * (wave >> 1) + 2 * (wave & 1), c1 = CPR / 2 + c0;
                unsigned char* P1 = lds + PSTR;
                *reinterpret_cast<uint4*>(P1 + choff(row0 + q, q, c0)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                *reinterpret_cast<uint2*>(P1 + choff(row0 + q, q, c0 + 1)) = make_uint2(pl[4], pl[5]);
                *reinterpret_cast<uint4*>(P1 + choff(row0 + q, q, c1)) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                *reinterpret_cast<uint2*>(P1 + choff(row0 + q, q, c1 + 1)) = make_uint2(pv[4], pv[5]);
            }
        }
    };
    // offsets of this lane's four channels ch .. ch + 3 of pixel row `row` (image-relative key for the swizzle): the f16 quad,
    // its lo8 word, its e4m3(x) word
    auto offs = [&](int row_abs, int key, int ch, int& off, int& off_lo, int& off_hi) {
        const int c_f = ch >> 3, c_lo = ch >> 4, c_hi = CPR / 2 + (ch >> 4);
        off = row_abs * RB + ((c_f & ~G::SWZ) << 4) + (((c_f ^ key) & G::SWZ) << 4) + (ch & 7) * 2;
        off_lo = PSTR + row_abs * RB + ((c_lo & ~G::SWZ) << 4) + (((c_lo ^ key) & G::SWZ) << 4) + (ch & 15);
        off_hi = PSTR + row_abs * RB + ((c_hi & ~G::SWZ) << 4) + (((c_hi ^ key) & G::SWZ) << 4) + (ch & 15);
    };
    int g = 0;                                                  // running block count: its parity picks the bias buffer
    for (;;) {
        __syncthreads();                                        // A
        const bool has_next = t + stride < n_boards;
      for (int blk = 0; blk < NB; ++blk, ++g) {
        const void* w1p = ch.w1[blk];
        const void* w2p = ch.w2[blk];
        const c8k::Filter flt1 = c8k::make_filter<C>(w1p, wave, lane), flt2 = c8k::make_filter<C>(w2p, wave, lane);
        const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF) + (g & 1) * 2 * C;
        const float* bias2 = bias1 + C;
        // c6: the exponents of the images the two convolutions read and of the one the block writes (127: a c8 image)
        const int* ints1 = reinterpret_cast<const int*>(reinterpret_cast<const uint4*>(w1p) + c8k::Geo<C>::MAIN_U4 + c8k::Geo<C>::C8_U4);
        const int* ints2 = reinterpret_cast<const int*>(reinterpret_cast<const uint4*>(w2p) + c8k::Geo<C>::MAIN_U4 + c8k::Geo<C>::C8_U4);
        const int k_x = XF ? __builtin_amdgcn_readfirstlane(ints1[2]) : 0;
        const int k_y = YF ? __builtin_amdgcn_readfirstlane(ints2[2]) : 0;
        const int k_out = YF ? __builtin_amdgcn_readfirstlane(ints2[3]) : CZ_C6_OUT_C8;
        float* yf = blk == NB - 1 ? yf_last : nullptr;          // fp32 output: the chain's last block only
        f32x16 acc[NT];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(bias1 + wave * 32 + g * 8 + kb * 4);
#pragma unroll
            for (int p = 0; p < NT; ++p) {
                acc[p][g * 4 + 0] = bv.x; acc[p][g * 4 + 1] = bv.y; acc[p][g * 4 + 2] = bv.z; acc[p][g * 4 + 3] = bv.w;
            }
        }
        __builtin_amdgcn_s_setprio(3);
        c8k::kloop<NT, c8k::NoShadow, 0, false, C, XF>(lds, c8k::Image{0, ip::ROW_Z, PSTR}, flt1, lane, acc, 127 + k_x - cf8::X_LO_SHIFT, 127 + k_x);
        __builtin_amdgcn_s_setprio(0);
        int ln2 = ln, kb2 = kb;
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 1: relu(acc) -> the operand triple (format YF) -> Y; the freed accumulators restart at b2 + skip (this lane's
        // own elements of X, format XF)
        if (YF) write_c6(acc, ip::ROW_Y, k_y, ln2, kb2);
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            const int row = q < 90 ? q : 89;
            rb8::f32x32 xl;
            if (XF) {                                           // this lane's elements of the x_lo piece of its 32-channel block
                const int c0 = 4 * (wave >> 1) + 2 * (wave & 1);
                const uint4 hd4 = *reinterpret_cast<const uint4*>(lds + PSTR + choff(row, row, c0));
                const uint2 tl2 = *reinterpret_cast<const uint2*>(lds + PSTR + choff(row, row, c0 + 1));
                const uint32_t wv[7] = {hd4.x, hd4.y, hd4.z, hd4.w, tl2.x, tl2.y, 0u};
                const uint32_t sh6 = (uint32_t)kb2 * 6u;       // (the upper lane half wants the odd elements: see k_resblock_c8)
                rb8::u32x6 pc;
#pragma unroll
                for (int w = 0; w < 6; ++w) pc[w] = __builtin_amdgcn_alignbit(wv[w + 1], wv[w], sh6);
                xl = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(pc, __builtin_ldexpf(1.0f, k_x - cf8::X_LO_SHIFT));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wave * 32 + g * 8 + kb2 * 4;
                int ox, oxl, oxh, oy, oyl, oyh;
                offs(row, row, ch, ox, oxl, oxh);
                offs(ip::ROW_Y + row, row, ch, oy, oyl, oyh);
                if (!YF) {
                    float r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                    const cf8::Split4 o = cf8::split4(r);
                    if (q < 90) {
                        *reinterpret_cast<Quad<_Float16>*>(lds + oy) = o.hi;
                        *reinterpret_cast<uint32_t*>(lds + oyl) = o.l8;
                        *reinterpret_cast<uint32_t*>(lds + oyh) = o.h8;
                    }
                }
                const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
                float vv[4] = {bv.x, bv.y, bv.z, bv.w};
                if (XF) {
                    const Quad<_Float16> xq = *reinterpret_cast<const Quad<_Float16>*>(lds + ox);
#pragma unroll
                    for (int i = 0; i < 4; ++i) vv[i] += (float)xq.e[i] + xl[2 * (g * 4 + i)];
                } else {
                    cf8::add_pair4(vv, *reinterpret_cast<const Quad<_Float16>*>(lds + ox), *reinterpret_cast<const uint32_t*>(lds + oxl));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[p][g * 4 + i] = vv[i];
            }
        }
        __syncthreads();                                        // B: Y complete
        __builtin_amdgcn_s_setprio(3);
        c8k::kloop<NT, c8k::NoShadow, 0, false, C, YF>(lds, c8k::Image{ip::ROW_Y, ip::ROW_Z, PSTR}, flt2, lane, acc,
                                                       127 + k_y - cf8::X_LO_SHIFT, 127 + k_y);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("" : "+v"(ln2), "+v"(kb2));
        // epilogue 2: relu(acc) -> the operand triple in place over the skip operand (X is dead: every lane consumed its skip
        // elements before barrier B), or fp32 straight to HBM for the last block of a tower
        if (YF && !yf && k_out != CZ_C6_OUT_C8) write_c6(acc, 0, k_out, ln2, kb2);
        else
#pragma unroll
        for (int p = 0; p < NT; ++p) {
            const int q = p * 32 + ln2;
            if (q < 90) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb2 * 4;
                    float r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                    if (yf) {
                        *reinterpret_cast<float4*>(yf + ((size_t)t * 90 + q) * C + ch) = make_float4(r[0], r[1], r[2], r[3]);
                    } else {
                        int ox, oxl, oxh;
                        offs(q, q, ch, ox, oxl, oxh);
                        const cf8::Split4 o = cf8::split4(r);
                        *reinterpret_cast<Quad<_Float16>*>(lds + ox) = o.hi;
                        *reinterpret_cast<uint32_t*>(lds + oxl) = o.l8;
                        *reinterpret_cast<uint32_t*>(lds + oxh) = o.h8;
                    }
                }
            }
        }
        __syncthreads();                                        // C: the result is in X
      }
        if (!has_next) break;
        t += stride;
    }
}

// ---- kernel 2c / c8 / four waves: k_resblock_ip_c8's chain on FOUR matrix waves of three channel tiles each, a pair of boards
// per workgroup (round 6) -------------------------------------------------------------------------------------------------------
// 192 filters are six channel tiles.  k_resblock_ip_c8 gives each of six matrix waves one tile: two SIMDs carry two matrix waves,
// two carry one -- the convolution takes as long as the SIMDs with two, the others idle half of it, and every wave reads every
// pixel fragment from LDS for ONE MFMA.  Here a workgroup holds two boards (A = rows [0, 90), B = rows [90, 180): the X and Y rows
// of k_resblock_ip_c8 -- its arithmetic restarts the accumulators at b2 + skip inside epilogue 1, so the intermediate activation
// can be written over a board's input IN PLACE) and has four matrix waves, one per SIMD with 512 registers: wave w takes board
// w >> 1, channel tiles 3 (w & 1) .. + 2, all three pixel tiles (c8k::kloop_ctw<3, 3>: 144 accumulators, a pixel fragment feeds
// three MFMAs).  36 (tile, pixel tile, board) units, nine per SIMD.  No copy waves: the four waves write the bias buffers and
// drain / refill the pair once per chain.  A c6 piece holds the 32 channels of ONE channel tile for a pixel and is written by
// lanes that traded pixel tiles: per channel tile a wave reads the skip elements of both tiles of a trade before it writes their
// pieces.  XF / YF as in k_resblock_ip_c8 (<0, 0> c8, <1, 1> c6, <0, 1> the tower's first c6 block; a chain runs one of them).  Per accumulator tile the same products in the same order and the same
// epilogue arithmetic as k_resblock_ip_c8: bit-identical.
constexpr int IP4_EXIT_PAIRS = 2, IP4_EXIT_HEADS = 3;
#ifndef CZ_IP4_PROBE        // timing experiments (wrong results; 0 in the product): bit 0 = no epilogue 1, bit 1 = no epilogue 2
#define CZ_IP4_PROBE 0
#endif
// MIX (with <1, 1>): the chain starts the tower -- its block 0 reads the input layer's c8 image (first filter c8-packed: CZ_F16C86)
template <int C, int XF, int YF, bool MIX = false>
__global__ __launch_bounds__(256, 1) void k_resblock_ip4_c8(
    const _Float16* __restrict__ xh, const unsigned char* __restrict__ xc, ip::Chain ch, _Float16* __restrict__ yh,
    unsigned char* __restrict__ yc, float* __restrict__ yf_last, int n_boards, const int32_t* __restrict__ n_dev,
    int exit_mode, HeadArgs hd)
{
    typedef Geom<C, 1, 2> G;
    constexpr int RB = G::RB, CPR = G::CPR, NT = 3, CTW = G::CT / 2, NTHR = 256;
    static_assert(G::CT == 2 * CTW, "two waves of CTW channel tiles per board");
    constexpr int PSTR = ip::ROWS * RB;                         // bytes per operand part
    constexpr int BIAS_OFF = 2 * PSTR;
    constexpr int CHUNKS = 180 * CPR, LITER = (CHUNKS + NTHR - 1) / NTHR;
    // exit_mode (the chain's LAST block, 128 filters: cz_tower's exits): 0 = the operand pair / fp32 (yf_last); IP4_EXIT_PAIRS =
    // (hi, lo) fp16 pairs [n][90][C] to yh and yc; IP4_EXIT_HEADS = the six 1 x 1 head features (hd).  Both stage relu(acc) as fp32
    // in the board's dead image ([pixel][64 channels] per part, 16-byte chunks swizzled by the pixel) and apply k_tower's exit
    // arithmetic, item for item: bit-identical to k_tower's exits.
    constexpr int HW_OFF = BIAS_OFF + 2 * 2 * C * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HW_OFF + (C == 128 ? 6 * C * 4 : 0)];   // bias[2 buffers][2 convolutions][C] | head filters
    const int NB = ch.n;
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pairs = (n_boards + 1) / 2;
    int t = blockIdx.x;
    if (t >= n_pairs) return;
    const int stride = gridDim.x;
    typedef c8k::u32x4 u4;
    // 16-byte chunk `chunk` of pixel row `key` of board `bd` (inside a part): the swizzle key is the board-relative row
    auto choff = [&](int bd, int key, int chunk) {
        return (bd * 90 + key) * RB + ((chunk & ~G::SWZ) << 4) + (((chunk ^ key) & G::SWZ) << 4);
    };
    auto chunk_off = [&](int i) {                               // chunk i of the pair
        const int row = i / CPR, c = i - row * CPR;
        return choff(row >= 90 ? 1 : 0, row >= 90 ? row - 90 : row, c);
    };
    auto fill = [&](int pr) __attribute__((always_inline)) {    // HBM -> images (a missing second board: zeros)
        const int have = (n_boards - 2 * pr < 2 ? n_boards - 2 * pr : 2) * 90 * CPR;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const u4* src = part ? reinterpret_cast<const u4*>(xc + (size_t)2 * pr * 90 * 2 * C)
                                 : reinterpret_cast<const u4*>(xh + (size_t)2 * pr * 90 * C);
#pragma unroll
            for (int it0 = 0; it0 < LITER; it0 += 9) {
                u4 v[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int i = (it0 + j) * NTHR + tid;
                    v[j] = u4{0u, 0u, 0u, 0u};
                    if (it0 + j < LITER && i < have) v[j] = src[i];
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int i = (it0 + j) * NTHR + tid;
                    if (it0 + j < LITER && i < CHUNKS) *reinterpret_cast<u4*>(lds + part * PSTR + chunk_off(i)) = v[j];
                }
            }
        }
    };
    auto drain = [&](int pr) __attribute__((always_inline)) {
        const int have = (n_boards - 2 * pr < 2 ? n_boards - 2 * pr : 2) * 90 * CPR;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            u4* dst = part ? reinterpret_cast<u4*>(yc + (size_t)2 * pr * 90 * 2 * C) : reinterpret_cast<u4*>(yh + (size_t)2 * pr * 90 * C);
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int i = it * NTHR + tid;
                if (i < have) dst[i] = *reinterpret_cast<const u4*>(lds + part * PSTR + chunk_off(i));
            }
        }
    };
    auto write_bias = [&](int g) {                              // biases of running block g (block g % NB) into buffer g & 1
        const int blk = g % NB;
        float* dst = reinterpret_cast<float*>(lds + BIAS_OFF) + (g & 1) * 2 * C;
        for (int i = tid; i < C; i += NTHR) {
            dst[i] = ch.b1[blk][i];
            dst[C + i] = ch.b2[blk][i];
        }
    };
    for (int i = tid; i < 16 * CPR; i += NTHR) {                // the shared zero rows, both parts
        *reinterpret_cast<u4*>(lds + ip::ROW_Z * RB + i * 16) = u4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u4*>(lds + PSTR + ip::ROW_Z * RB + i * 16) = u4{0u, 0u, 0u, 0u};
    }
    fill(t);
    write_bias(0);
    write_bias(1);
    if (C == 128 && exit_mode == IP4_EXIT_HEADS)
        for (int i = tid; i < 6 * C; i += NTHR) reinterpret_cast<float*>(lds + HW_OFF)[i] = hd.w[i];

    const int kb = lane >> 5, ln = lane & 31;
    const int bd = wave >> 1, tile0 = CTW * (wave & 1);         // this wave's board and first channel tile
    // this lane's four channels chn .. chn + 3 of pixel row `key` of its board: the f16 quad, its lo8 word, its e4m3(x) word
    auto offs = [&](int key, int chn, int& off, int& off_lo, int& off_hi) {
        off = choff(bd, key, chn >> 3) + (chn & 7) * 2;
        off_lo = PSTR + choff(bd, key, chn >> 4) + (chn & 15);
        off_hi = PSTR + choff(bd, key, CPR / 2 + (chn >> 4)) + (chn & 15);
    };
    // one unit of a c6 image: channel tile tc, pixel tiles (0, 1) (pp = 0) or tile 2 with itself (pp = 1) of this wave's board.
    // relu(a*) -> f16 quads + the two bf6 pieces of a pixel's 32 channels (exponent k); a* keep relu(a*).
    auto write_c6_unit = [&](f32x16& aa, f32x16& ab, int tc, int pp, int k, int ln2, int kb2) __attribute__((always_inline)) {
        using rb8::u32x6;
        const float s_hi = __builtin_ldexpf(1.0f, k), s_lo = __builtin_ldexpf(1.0f, k - cf8::X_LO_SHIFT);
        f32x16 lo_a, lo_b;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && pp == 1) break;
            const int tt = pp == 0 ? h : 2;
            const int q = tt * 32 + ln2;
            const int key = q < 90 ? q : 89;
            f32x16& a = h ? ab : aa;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int chn = tc * 32 + gg * 8 + kb2 * 4;
                Quad<_Float16> hq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = a[gg * 4 + i] > 0.0f ? a[gg * 4 + i] : 0.0f;
                    hq.e[i] = (_Float16)r;
                    a[gg * 4 + i] = r;
                    (h ? lo_b : lo_a)[gg * 4 + i] = r - (float)hq.e[i];
                }
                if (q < 90) *reinterpret_cast<Quad<_Float16>*>(lds + choff(bd, key, chn >> 3) + (chn & 7) * 2) = hq;
            }
        }
        if (pp == 1) lo_b = lo_a;
        f32x16 av, bv, al, bl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(aa[r]), __float_as_uint(pp == 0 ? ab[r] : aa[r]), false, false);
            const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_a[r]), __float_as_uint(lo_b[r]), false, false);
            av[r] = __uint_as_float(sv[0]); bv[r] = __uint_as_float(sv[1]);
            al[r] = __uint_as_float(sl[0]); bl[r] = __uint_as_float(sl[1]);
        }
        const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
        const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
        const int q = pp == 0 ? kb2 * 32 + ln2 : 64 + ln2;
        if (pp == 0 || (kb2 == 0 && q < 90)) {
            const int c0 = 4 * (tc >> 1) + 2 * (tc & 1), c1 = CPR / 2 + c0;
            unsigned char* P1 = lds + PSTR;
            *reinterpret_cast<u4*>(P1 + choff(bd, q, c0)) = u4{pl[0], pl[1], pl[2], pl[3]};
            *reinterpret_cast<c8k::u32x2*>(P1 + choff(bd, q, c0 + 1)) = c8k::u32x2{pl[4], pl[5]};
            *reinterpret_cast<u4*>(P1 + choff(bd, q, c1)) = u4{pv[0], pv[1], pv[2], pv[3]};
            *reinterpret_cast<c8k::u32x2*>(P1 + choff(bd, q, c1 + 1)) = c8k::u32x2{pv[4], pv[5]};
        }
    };
    int g = 0;                                                  // running block count: its parity picks the bias buffer
    for (;;) {
        __syncthreads();                                        // A: the images hold pair t, the bias buffers are written
        for (int blk = 0; blk < NB; ++blk, ++g) {
            const void* w1p = ch.w1[blk];
            const void* w2p = ch.w2[blk];
            c8k::Filter flt1[CTW], flt2[CTW];
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                flt1[c] = c8k::make_filter<C>(w1p, tile0 + c, lane);
                flt2[c] = c8k::make_filter<C>(w2p, tile0 + c, lane);
            }
            const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF) + (g & 1) * 2 * C;
            const float* bias2 = bias1 + C;
            const int* ints1 = reinterpret_cast<const int*>(reinterpret_cast<const uint4*>(w1p) + c8k::Geo<C>::MAIN_U4 + c8k::Geo<C>::C8_U4);
            const int* ints2 = reinterpret_cast<const int*>(reinterpret_cast<const uint4*>(w2p) + c8k::Geo<C>::MAIN_U4 + c8k::Geo<C>::C8_U4);
            static_assert(!MIX || (XF == 1 && YF == 1), "a c6 chain whose first block reads a c8 image");
            const bool xf = MIX ? blk != 0 : XF != 0;           // the format of the image this block's first convolution reads
            const int k_x = xf ? __builtin_amdgcn_readfirstlane(ints1[2]) : 0;
            const int k_y = YF ? __builtin_amdgcn_readfirstlane(ints2[2]) : 0;
            const int k_out = YF ? __builtin_amdgcn_readfirstlane(ints2[3]) : CZ_C6_OUT_C8;
            float* yf = blk == NB - 1 ? yf_last : nullptr;      // fp32 output: the chain's last block only
            f32x16 acc[CTW * NT];
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias1 + (tile0 + c) * 32 + gg * 8 + kb * 4);
#pragma unroll
                    for (int p = 0; p < NT; ++p) {
                        acc[c * NT + p][gg * 4 + 0] = bv.x; acc[c * NT + p][gg * 4 + 1] = bv.y;
                        acc[c * NT + p][gg * 4 + 2] = bv.z; acc[c * NT + p][gg * 4 + 3] = bv.w;
                    }
                }
            const c8k::Image img{bd * 90, ip::ROW_Z, PSTR};
            __builtin_amdgcn_s_setprio(3);
            if (MIX && !xf) c8k::kloop_ctw<CTW, NT, C, 0>(lds, img, flt1, lane, acc, 127 - cf8::X_LO_SHIFT, 127);
            else c8k::kloop_ctw<CTW, NT, C, XF>(lds, img, flt1, lane, acc, 127 + k_x - cf8::X_LO_SHIFT, 127 + k_x);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();                                    // K1: both waves of a board have read its image
            int ln2 = ln, kb2 = kb;
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 1, in place: this lane's skip elements out of the image, relu(acc) in the operand format over them, the
            // freed accumulators restart at b2 + skip
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int tc = tile0 + c;
                if (CZ_IP4_PROBE & 1) continue;
                if (YF) {
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        f32x16 sk[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (h == 1 && pp == 1) break;
                            const int tt = pp == 0 ? h : 2;
                            const int q = tt * 32 + ln2;
                            const int key = q < 90 ? q : 89;
                            rb8::f32x32 xl;
                            if (xf) {                           // this lane's elements of the x_lo piece of the channel tile
                                const int c0 = 4 * (tc >> 1) + 2 * (tc & 1);
                                const u4 hd4 = *reinterpret_cast<const u4*>(lds + PSTR + choff(bd, key, c0));
                                const c8k::u32x2 tl2 = *reinterpret_cast<const c8k::u32x2*>(lds + PSTR + choff(bd, key, c0 + 1));
                                const uint32_t wv[7] = {hd4.x, hd4.y, hd4.z, hd4.w, tl2.x, tl2.y, 0u};
                                const uint32_t sh6 = (uint32_t)kb2 * 6u;
                                rb8::u32x6 pc;
#pragma unroll
                                for (int w = 0; w < 6; ++w) pc[w] = __builtin_amdgcn_alignbit(wv[w + 1], wv[w], sh6);
                                xl = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(pc, __builtin_ldexpf(1.0f, k_x - cf8::X_LO_SHIFT));
                            }
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                const int chn = tc * 32 + gg * 8 + kb2 * 4;
                                const float4 bv = *reinterpret_cast<const float4*>(bias2 + chn);
                                float vv[4] = {bv.x, bv.y, bv.z, bv.w};
                                if (xf) {
                                    const Quad<_Float16> xq = *reinterpret_cast<const Quad<_Float16>*>(lds + choff(bd, key, chn >> 3) + (chn & 7) * 2);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) vv[i] += (float)xq.e[i] + xl[2 * (gg * 4 + i)];
                                } else {                        // (the tower's first c6 block: the input layer's c8 image)
                                    int ox, oxl, oxh;
                                    offs(key, chn, ox, oxl, oxh);
                                    cf8::add_pair4(vv, *reinterpret_cast<const Quad<_Float16>*>(lds + ox), *reinterpret_cast<const uint32_t*>(lds + oxl));
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) sk[h][gg * 4 + i] = vv[i];
                            }
                        }
                        if (pp == 0) {
                            write_c6_unit(acc[c * NT + 0], acc[c * NT + 1], tc, 0, k_y, ln2, kb2);
                            acc[c * NT + 0] = sk[0];
                            acc[c * NT + 1] = sk[1];
                        } else {
                            write_c6_unit(acc[c * NT + 2], acc[c * NT + 2], tc, 1, k_y, ln2, kb2);
                            acc[c * NT + 2] = sk[0];
                        }
                    }
                } else {
                    static_assert(YF || !XF, "a c8 intermediate image behind a c6 input does not exist");
#pragma unroll
                    for (int p = 0; p < NT; ++p) {
                        const int q = p * 32 + ln2;
                        const int key = q < 90 ? q : 89;
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) {
                            const int chn = tc * 32 + gg * 8 + kb2 * 4;
                            int ox, oxl, oxh;
                            offs(key, chn, ox, oxl, oxh);
                            const float4 bv = *reinterpret_cast<const float4*>(bias2 + chn);
                            float vv[4] = {bv.x, bv.y, bv.z, bv.w};
                            cf8::add_pair4(vv, *reinterpret_cast<const Quad<_Float16>*>(lds + ox), *reinterpret_cast<const uint32_t*>(lds + oxl));
                            float r[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] = acc[c * NT + p][gg * 4 + i] > 0.0f ? acc[c * NT + p][gg * 4 + i] : 0.0f;
                            const cf8::Split4 o = cf8::split4(r);
                            if (q < 90) {
                                *reinterpret_cast<Quad<_Float16>*>(lds + ox) = o.hi;
                                *reinterpret_cast<uint32_t*>(lds + oxl) = o.l8;
                                *reinterpret_cast<uint32_t*>(lds + oxh) = o.h8;
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[c * NT + p][gg * 4 + i] = vv[i];
                        }
                    }
                }
            }
            __syncthreads();                                    // B: the images hold the intermediate activation; block g's biases are consumed
            if (NB > 1) write_bias(g + 2);
            __builtin_amdgcn_s_setprio(3);
            c8k::kloop_ctw<CTW, NT, C, YF>(lds, img, flt2, lane, acc, 127 + k_y - cf8::X_LO_SHIFT, 127 + k_y);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();                                    // K2: both waves of a board have read it
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 2: relu(acc) -> the operand triple over the image, or fp32 straight to HBM for the last block of a tower
            const int ex = blk == NB - 1 ? exit_mode : 0;
            // byte offset of channels chn .. chn + 3 (fp32) of pixel q in the board's staging (128 filters)
            auto stg = [&](int q, int chn) {
                return (chn >> 6) * PSTR + (bd * 90 + q) * RB + (((((chn & 63) >> 2)) ^ (q & 15)) << 4);
            };
            if (CZ_IP4_PROBE & 2) {
            } else if (C == 128 && ex != 0) {
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int p = 0; p < NT; ++p) {
                        const int q = p * 32 + ln2;
                        if (q < 90) {
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                const int chn = (tile0 + c) * 32 + gg * 8 + kb2 * 4;
                                float r[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) r[i] = acc[c * NT + p][gg * 4 + i] > 0.0f ? acc[c * NT + p][gg * 4 + i] : 0.0f;
                                *reinterpret_cast<float4*>(lds + stg(q, chn)) = make_float4(r[0], r[1], r[2], r[3]);
                            }
                        }
                    }
            } else
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int tc = tile0 + c;
                if (YF && !yf && k_out != CZ_C6_OUT_C8) {
                    write_c6_unit(acc[c * NT + 0], acc[c * NT + 1], tc, 0, k_out, ln2, kb2);
                    write_c6_unit(acc[c * NT + 2], acc[c * NT + 2], tc, 1, k_out, ln2, kb2);
                } else {
#pragma unroll
                    for (int p = 0; p < NT; ++p) {
                        const int q = p * 32 + ln2;
                        const int board = 2 * t + bd;
                        if (q < 90) {
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                const int chn = tc * 32 + gg * 8 + kb2 * 4;
                                float r[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) r[i] = acc[c * NT + p][gg * 4 + i] > 0.0f ? acc[c * NT + p][gg * 4 + i] : 0.0f;
                                if (yf) {
                                    if (board < n_boards)
                                        *reinterpret_cast<float4*>(yf + ((size_t)board * 90 + q) * C + chn) = make_float4(r[0], r[1], r[2], r[3]);
                                } else {
                                    int ox, oxl, oxh;
                                    offs(q, chn, ox, oxl, oxh);
                                    const cf8::Split4 o = cf8::split4(r);
                                    *reinterpret_cast<Quad<_Float16>*>(lds + ox) = o.hi;
                                    *reinterpret_cast<uint32_t*>(lds + oxl) = o.l8;
                                    *reinterpret_cast<uint32_t*>(lds + oxh) = o.h8;
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();                                    // C: the result is in the images
            if (C == 128 && ex != 0) {
                // the exit of board 2 t + bd by its two waves: item i = (pixel i >> 2, 32-channel block i & 3), as k_tower's copy waves
                const int board = 2 * t + bd;
                struct alignas(16) H8 { Quad<_Float16> a, b; };
                const float* hwl = reinterpret_cast<const float*>(lds + HW_OFF);
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int i = it * 128 + (wave & 1) * 64 + lane;
                    if (i >= 90 * 4) continue;
                    const int qq = i >> 2, b32 = i & 3;
                    if (ex == IP4_EXIT_PAIRS) {
                        if (board >= n_boards) continue;
                        const size_t ebase = (size_t)board * 90 * C;
                        _Float16* yl = reinterpret_cast<_Float16*>(yc);
#pragma unroll
                        for (int k8 = 0; k8 < 4; ++k8) {
                            const float4 f0 = *reinterpret_cast<const float4*>(lds + stg(qq, b32 * 32 + 8 * k8));
                            const float4 f1 = *reinterpret_cast<const float4*>(lds + stg(qq, b32 * 32 + 8 * k8 + 4));
                            const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                            H8 hi, lo;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                hi.a.e[j] = (_Float16)r[j];
                                hi.b.e[j] = (_Float16)r[4 + j];
                                lo.a.e[j] = (_Float16)(r[j] - (float)hi.a.e[j]);
                                lo.b.e[j] = (_Float16)(r[4 + j] - (float)hi.b.e[j]);
                            }
                            reinterpret_cast<u4*>(yh + ebase)[qq * 16 + b32 * 4 + k8] = __builtin_bit_cast(u4, hi);
                            reinterpret_cast<u4*>(yl + ebase)[qq * 16 + b32 * 4 + k8] = __builtin_bit_cast(u4, lo);
                        }
                    } else {
                        float a6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int k4 = 0; k4 < 8; ++k4) {
                            const float4 f = *reinterpret_cast<const float4*>(lds + stg(qq, b32 * 32 + 4 * k4));
#pragma unroll
                            for (int o = 0; o < 6; ++o) {
                                const float4 w = *reinterpret_cast<const float4*>(hwl + o * C + b32 * 32 + 4 * k4);
                                a6[o] += f.x * w.x; a6[o] += f.y * w.y; a6[o] += f.z * w.z; a6[o] += f.w * w.w;
                            }
                        }
#pragma unroll
                        for (int o = 0; o < 6; ++o) {              // the four lanes of the pixel: (a0 + a1) + (a2 + a3) on every lane
                            a6[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a6[o]), 0xB1, 0xF, 0xF, true));
                            a6[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a6[o]), 0x4E, 0xF, 0xF, true));
                        }
                        if (board >= n_boards) continue;
#pragma unroll
                        for (int o = 0; o < 6; ++o)
                            if ((o & 3) == b32) {                   // lane b32 writes outputs b32 and b32 + 4
                                float hv = a6[o] + hd.b[o];
                                hv = hv > 0.0f ? hv : 0.0f;
                                if (o < hd.n_pol) hd.pol[(size_t)board * (hd.n_pol * 90) + o * 90 + qq] = hv;
                                else hd.val[(size_t)board * ((6 - hd.n_pol) * 90) + (o - hd.n_pol) * 90 + qq] = hv;
                            }
                    }
                }
            }
        }
        if (!yf_last && exit_mode == 0) drain(t);
        t += stride;
        if (t >= n_pairs) break;
        __syncthreads();                                        // (drain has read the images)
        fill(t);
    }
}

// ---- kernel 3: the input convolution (5x5, 14 or 28 feature planes -> C channels) ----------------------------------
// Reference: Conv2D(F, 5, padding="same") -> BatchNorm -> ReLU on the state_to_planes input (agent/model.py:36-39).
// The planes arrive exactly as the search kernel writes them ([in_planes][10][9] per board, values 0 / 1, any of
// fp32 / fp16 / bf16 / u8) and are transposed into a channels-last LDS image ([pixel][16 or 32 channels], zero padded)
// while being staged.  0 / 1 are exact in bf16, so only the WEIGHTS are split: two MFMAs per product in split mode.
// One K-step per tap and 16 input channels; output written as the (hi, lo) operand pair of the residual tower.
template <typename PT> __device__ __forceinline__ float plane_to_f(PT v) { return (float)v; }

template <typename E, typename PT, int C, int IC16, int P, int PARTS, bool C8 = false>
__global__ __launch_bounds__(C / 32 * 64, 1) void k_input_conv(
    const PT* __restrict__ planes, const E* __restrict__ wp, const float* __restrict__ bias, E* __restrict__ yh,
    E* __restrict__ yl, int n_boards, int in_planes, int relu, const int32_t* __restrict__ rows,
    const int32_t* __restrict__ n_dev)
{
    if (n_dev) {                                    // compact queue: the board count lives on the device
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
        if ((int)blockIdx.x * P >= n_boards) return;
    }
    typedef typename Mfma<E>::V8 V8;
    constexpr int RBI = IC16 * 32;                  // bytes per pixel row of the input image
    // 16 zero rows and a swizzle, as in Geom: the 256-byte bank row holds RPB = 8 (or 4) of these short pixel rows, so
    // the 16 consecutive rows of a ds_read_b128 lane group would share 8 (4) slots; chunk ^= (row / RPB) gives each
    // of them its own 16-byte slot (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.60 here in round 1)
    constexpr int CPRI = IC16 * 2, RPB = 16 / CPRI;
    constexpr int ZROW = (P * 90 + 15) / 16 * 16;
    constexpr int IMG = (ZROW + 16) * RBI;
    constexpr int NT = P * 3, CT = C / 32, NTHR = CT * 64;
    constexpr int NX = NT * IC16;                   // operand reads per tap
    constexpr int NM = NX * PARTS;                  // MFMAs per tap
    constexpr int W_STEP = CT * 64;                 // uint4 per (tap, 16-channel group)
    constexpr int W_PART = (25 + 3) * IC16 * W_STEP;
    __shared__ __attribute__((aligned(16))) unsigned char img[IMG];
    __shared__ __attribute__((aligned(16))) unsigned char stage[PARTS * 90 * C * 2];    // one board of output

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * P;
    for (int i = tid; i < IMG / 16; i += NTHR) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    {
        const int per_board = in_planes * 90;
        const int nb = n_boards - n0 < P ? n_boards - n0 : P;
        // the image is all zero: only the occupied squares are written (a position has at most 32 of 1260 planes' bits set)
        auto put = [&](int bb, int c, int pix, float v) {
            const int row = bb * 90 + pix;
            *reinterpret_cast<E*>(img + row * RBI + (((c >> 3) ^ ((row / RPB) & (CPRI - 1))) << 4) + (c & 7) * 2) = (E)v;
        };
        if (sizeof(PT) == 1 && (per_board & 3) == 0) {
            // byte planes (what the search kernel writes for this network): four squares per load
            const int wpb = per_board >> 2;
            for (int i = tid; i < nb * wpb; i += NTHR) {
                const int bb = i / wpb, w = i - bb * wpb;
                // (compact queue: board n0 + bb of the batch is queue slot rows[n0 + bb])
                const uint32_t word = reinterpret_cast<const uint32_t*>(
                    reinterpret_cast<const unsigned char*>(planes) + (size_t)(rows ? rows[n0 + bb] : n0 + bb) * per_board)[w];
                if (word == 0u) continue;
                int c = (w * 4) / 90, pix = w * 4 - c * 90;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned b = (word >> (8 * k)) & 0xFFu;
                    if (b) put(bb, c, pix, (float)b);
                    if (++pix == 90) { pix = 0; ++c; }
                }
            }
        } else {
            for (int i = tid; i < nb * per_board; i += NTHR) {
                const int bb = i / per_board, r = i - bb * per_board;
                const PT* src = planes + (size_t)(rows ? rows[n0 + bb] : n0 + bb) * per_board - (size_t)bb * per_board;
                const int c = r / 90, pix = r - c * 90;
                const float v = plane_to_f(src[i]);
                if (v != 0.0f) put(bb, c, pix, v);
            }
        }
    }
    __syncthreads();

    const int kb = lane >> 5, ln = lane & 31;
    int qy[3], qx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = t * 32 + ln;
        qy[t] = q < 90 ? q / 9 : 100;
        qx[t] = q - (q / 9) * 9;
    }
    auto tap_off = [&](int tap, int p, int g) {     // byte offset of this lane's 16 bytes for (tap, 16-channel group g)
        const int ky = tap / 5;
        const int dy = ky - 2, dx = tap - ky * 5 - 2;
        const int t = p % 3;
        const bool ok = (unsigned)(qy[t] + dy) < 10u && (unsigned)(qx[t] + dx) < 9u;
        const int nominal = (p / 3) * 90 + t * 32 + ln + dy * 9 + dx;
        const int row = ok ? nominal : ZROW + (nominal & 15);
        return row * RBI + ((((g << 1) | kb) ^ ((row / RPB) & (CPRI - 1))) << 4);
    };
    const uint4* wq = reinterpret_cast<const uint4*>(wp) + wave * 64 + lane;
    auto load_w = [&](int tap, int g, int part) {
        return __builtin_bit_cast(V8, wq[(size_t)part * W_PART + (size_t)(tap * IC16 + g) * W_STEP]);
    };
    V8 wf[4][IC16][PARTS];
    V8 px[2][NX];
    f32x16 acc[NT];
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int g = 0; g < IC16; ++g)
#pragma unroll
            for (int part = 0; part < PARTS; ++part) wf[s][g][part] = load_w(s, g, part);
#pragma unroll
    for (int i = 0; i < NX; ++i)
        px[0][i] = __builtin_bit_cast(V8, *reinterpret_cast<const uint4*>(img + tap_off(0, i % NT, i / NT)));

    auto tap_body = [&](int tap, const int ring, const int buf) {
        const int tn = tap < 24 ? tap + 1 : 24;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int xi = i % NX, part = i / NX;          // all operands with w_hi first, then with w_lo
            acc[xi % NT] = Mfma<E>::mma(wf[ring][xi / NT][part], px[buf][xi], acc[xi % NT]);
            if (i < NX)
                px[buf ^ 1][i] = __builtin_bit_cast(
                    V8, *reinterpret_cast<const uint4*>(img + tap_off(tn, i % NT, i / NT)));
            if (i >= NM - IC16 * PARTS) {
                const int j = i - (NM - IC16 * PARTS);
                wf[(ring + 3) & 3][j / PARTS][j % PARTS] = load_w(tap + 3, j / PARTS, j % PARTS);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (int t4 = 0; t4 < 24; t4 += 4) {
        tap_body(t4 + 0, 0, 0);
        tap_body(t4 + 1, 1, 1);
        tap_body(t4 + 2, 2, 0);
        tap_body(t4 + 3, 3, 1);
    }
    tap_body(24, 0, 0);

    // ---- epilogue: + bias, ReLU, split; one board at a time through an LDS staging image so that HBM sees whole
    // pixel rows (16 bytes per lane, 1 KiB per wave instruction).  Round 2 stored the accumulators straight from the
    // MFMA layout -- 8 bytes per lane, 32 different 256-byte rows per instruction -- and the layer ran at 1.5 TB/s:
    // 94 M sixteen-byte write requests per launch are a request-rate limit, not a bandwidth one.
    typedef Geom<C, 1, PARTS> GS;                   // staging geometry: [90 pixels][C] per part, chunks swizzled by row
    constexpr int SROWB = C * 2, SPART = 90 * SROWB;
    f32x4 bq[4];                                    // this wave's biases, fetched once (not once per 8-byte store: see k_resblock)
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(bias + wave * 32 + g * 8 + kb * 4);
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[g]));
#pragma unroll
    for (int bb = 0; bb < P; ++bb) {
        if (bb > 0) __syncthreads();                // the previous board has left the staging image
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int p = bb * 3 + t;
            const int q = t * 32 + ln;
            if (q >= 90) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = wave * 32 + g * 8 + kb * 4;
                const f32x4 bv = bq[g];
                float v[4] = {acc[p][g * 4 + 0] + bv[0], acc[p][g * 4 + 1] + bv[1], acc[p][g * 4 + 2] + bv[2],
                              acc[p][g * 4 + 3] + bv[3]};
                const int off = q * SROWB + (((ch >> 3) ^ (q & GS::SWZ)) << 4) + (ch & 7) * 2;
                if constexpr (C8) {                 // the operand pair of the c8 arithmetic (k_conv3x3_c8): f16 row, then the c8 row
                    static_assert(!C8 || (PARTS == 2 && sizeof(E) == 2), "c8 output: fp16 operand pairs");
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (relu) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                    const cf8::Split4 o = cf8::split4(v);
                    *reinterpret_cast<Quad<_Float16>*>(stage + off) = o.hi;
                    unsigned char* crow = stage + SPART + q * SROWB + (ch & 15);
                    *reinterpret_cast<uint32_t*>(crow + (((ch >> 4) ^ (q & GS::SWZ)) << 4)) = o.l8;
                    *reinterpret_cast<uint32_t*>(crow + (((GS::CPR / 2 + (ch >> 4)) ^ (q & GS::SWZ)) << 4)) = o.h8;
                    continue;
                }
                Quad<E> hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (relu) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                    hi.e[i] = (E)v[i];
                    lo.e[i] = (E)(v[i] - (float)hi.e[i]);
                }
                *reinterpret_cast<Quad<E>*>(stage + off) = hi;
                if (PARTS == 2) *reinterpret_cast<Quad<E>*>(stage + SPART + off) = lo;
            }
        }
        __syncthreads();
        const int n = n0 + bb;
        if (n < n_boards) {
#pragma unroll
            for (int part = 0; part < PARTS; ++part) {
                uint4* dst = reinterpret_cast<uint4*>((part ? yl : yh) + (size_t)n * 90 * C);
                for (int i = tid; i < 90 * GS::CPR; i += NTHR) {
                    const int row = i / GS::CPR, chn = i - row * GS::CPR;
                    dst[i] = *reinterpret_cast<const uint4*>(stage + part * SPART + row * SROWB +
                                                             ((chn ^ (row & GS::SWZ)) << 4));
                }
            }
        }
    }
}

// ---- fp32 activation -> (hi, lo) operand pair, with bias and ReLU (after the 5x5 input convolution) ---------------
template <typename E, int PARTS>
__global__ __launch_bounds__(256) void k_split_bias_act(const float* __restrict__ x, const float* __restrict__ bias,
                                                       E* __restrict__ yh, E* __restrict__ yl, size_t nquad,
                                                       int cquad, int relu)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nquad; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        float v[4] = {a.x, a.y, a.z, a.w};
        if (bias) {
            const float4 b = reinterpret_cast<const float4*>(bias)[i % (size_t)cquad];
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        Quad<E> hi, lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (relu) v[k] = v[k] > 0.0f ? v[k] : 0.0f;
            hi.e[k] = (E)v[k];
            lo.e[k] = (E)(v[k] - (float)hi.e[k]);
        }
        reinterpret_cast<Quad<E>*>(yh)[i] = hi;
        if (PARTS == 2) reinterpret_cast<Quad<E>*>(yl)[i] = lo;
    }
}

template <typename E, int C, int P, int PARTS, int MINW = 1>
int launch_conv(const void* xh, const void* xl, const void* wp, const float* bias, const void* sh, const void* sl,
                void* yh, void* yl, float* yf, int n_boards, int relu, hipStream_t st)
{
    const unsigned blocks = (unsigned)((n_boards + P - 1) / P);
    hipLaunchKernelGGL((k_conv3x3<E, C, P, PARTS, MINW>), dim3(blocks), dim3(C / 32 * 64), 0, st, (const E*)xh,
                       (const E*)xl, (const E*)wp, bias, (const E*)sh, (const E*)sl, (E*)yh, (E*)yl, yf, n_boards,
                       relu);
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

template <typename E>
int dispatch_conv(int channels, int parts, const void* xh, const void* xl, const void* wp, const float* bias,
                  const void* sh, const void* sl, void* yh, void* yl, float* yf, int n, int relu, hipStream_t st)
{
#define CZ_CONV_ARGS xh, xl, wp, bias, sh, sl, yh, yl, yf, n, relu, st
    if (channels == 128 && parts == 2) return launch_conv<E, 128, 2, 2, 1>(CZ_CONV_ARGS);
    if (channels == 128 && parts == 1) return launch_conv<E, 128, 2, 1, 2>(CZ_CONV_ARGS);   // 2 workgroups / CU
    if (channels == 192 && parts == 2) return launch_conv<E, 192, 1, 2, 1>(CZ_CONV_ARGS);
    if (channels == 192 && parts == 1) return launch_conv<E, 192, 2, 1, 1>(CZ_CONV_ARGS);
    if (channels == 256 && parts == 2) return launch_conv<E, 256, 1, 2>(xh, xl, wp, bias, sh, sl, yh, yl, yf, n, relu, st);
    if (channels == 256 && parts == 1) return launch_conv<E, 256, 2, 1>(xh, xl, wp, bias, sh, sl, yh, yl, yf, n, relu, st);
    if (channels == 32 && parts == 2) return launch_conv<E, 32, 2, 2>(xh, xl, wp, bias, sh, sl, yh, yl, yf, n, relu, st);
    if (channels == 32 && parts == 1) return launch_conv<E, 32, 4, 1>(xh, xl, wp, bias, sh, sl, yh, yl, yf, n, relu, st);
    return CZ_ERR_ARG;
}

// round-to-nearest-even conversions on the host (pack_weights)
inline uint16_t f32_to_bf16_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_bits_to_f32(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline uint16_t f32_to_f16_bits(float f)
{
    const _Float16 h = (_Float16)f;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
inline float f16_bits_to_f32(uint16_t b)
{
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

}  // namespace

#ifdef CZ_RB_STAMPS
extern "C" int cz_debug_rb_knob(int v)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rb_knob), &v, sizeof(int)) == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}
extern "C" int cz_debug_rb_stamps(long long* out32_host)
{
    return hipMemcpyFromSymbol(out32_host, HIP_SYMBOL(g_rb_stamps), sizeof(long long) * 32) == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}
#endif

extern "C" size_t cz_conv3x3_packed_elems(int channels, int parts)
{
    if (channels <= 0 || channels % 32 != 0 || parts < 1 || parts > 2) return 0;
    return (size_t)parts * (size_t)(9 * (channels / 16) + W_PAD_STEPS) * (size_t)(channels / 32) * 64 * 8;
}

extern "C" int cz_conv3x3_pack_weights(const float* w_oihw, int channels, int dtype, int parts, void* out_host)
{
    if (!w_oihw || !out_host || channels <= 0 || channels % 32 != 0 || parts < 1 || parts > 2 ||
        (dtype != CZ_BF16 && dtype != CZ_F16)) {
        czi_set_error("cz_conv3x3_pack_weights: bad argument (channels % 32 == 0, parts 1|2, dtype bf16|f16)");
        return CZ_ERR_ARG;
    }
    const int C = channels, KK = C / 16, CT = C / 32;
    const size_t part_elems = (size_t)(9 * KK + W_PAD_STEPS) * CT * 64 * 8;
    uint16_t* out = (uint16_t*)out_host;
    memset(out, 0, part_elems * parts * sizeof(uint16_t));
    for (int tap = 0; tap < 9; ++tap)
        for (int kk = 0; kk < KK; ++kk)
            for (int ct = 0; ct < CT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = ct * 32 + (lane & 31);
                        const int c = kk * 16 + (lane >> 5) * 8 + j;
                        const float w = w_oihw[((size_t)o * C + c) * 9 + tap];
                        const size_t idx = ((((size_t)tap * KK + kk) * CT + ct) * 64 + lane) * 8 + j;
                        if (dtype == CZ_BF16) {
                            const uint16_t hi = f32_to_bf16_bits(w);
                            out[idx] = hi;
                            if (parts == 2) out[part_elems + idx] = f32_to_bf16_bits(w - bf16_bits_to_f32(hi));
                        } else {
                            const uint16_t hi = f32_to_f16_bits(w);
                            out[idx] = hi;
                            if (parts == 2) out[part_elems + idx] = f32_to_f16_bits(w - f16_bits_to_f32(hi));
                        }
                    }
    return CZ_OK;
}

// ---- c8 arithmetic (k_conv3x3_c8, k_resblock<C8>): packing and launch ------------------------------------------------------------
namespace {
// float -> OCP e4m3 (bias 7, no infinities, 0x7f = NaN), round to nearest even, saturating at 448
inline uint8_t f32_to_e4m3_bits(float f)
{
    uint32_t fb;
    memcpy(&fb, &f, 4);
    const uint8_t sign = (uint8_t)((fb >> 24) & 0x80);
    float a = fabsf(f);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0f) return sign | 0x7e;
    if (a < ldexpf(1.0f, -10)) return sign;                        // below half the smallest subnormal (2^-9): 0 (tie -> even = 0)
    int e;
    frexpf(a, &e);
    e -= 1;                                                        // a = m * 2^e, m in [1, 2)
    if (e < -6) {                                                  // subnormal: multiples of 2^-9
        const int q = (int)nearbyintf(ldexpf(a, 9));               // 0 .. 8 (8 = the smallest normal, code 0x08)
        return sign | (uint8_t)q;
    }
    int m = (int)nearbyintf(ldexpf(a, 3 - e)) - 8;                 // 0 .. 8
    if (m == 8) { m = 0; e += 1; }
    if (e > 8 || (e == 8 && m > 6)) return sign | 0x7e;
    return sign | (uint8_t)(((e + 7) << 3) | m);
}
// fp32 -> bf6 (e3m2, bias 3, no inf / nan): round to nearest even, saturating at 28 -- v_cvt_scalef32_2xpk16_bf6_f32's rule
inline uint8_t f32_to_bf6_bits(float f)
{
    const uint8_t sgn = std::signbit(f) ? 0x20 : 0;
    float a = fabsf(f);
    if (!(a == a)) return sgn;                                      // (NaN: no encoding; filters never hold one)
    if (a >= 28.0f) return sgn | 0x1F;
    int e;
    frexpf(a, &e);                                                  // a = m 2^e, m in [0.5, 1)
    int ex = e - 1;                                                 // a in [2^ex, 2^(ex + 1))
    if (ex < -2) ex = -2;                                           // subnormals share the smallest normal's step 2^-4
    const float step = ldexpf(1.0f, ex - 2);
    const float qf = nearbyintf(a / step);                          // (default rounding mode: to nearest even)
    const int qi = (int)qf;                                         // 0 .. 8 (8: carries into the next binade)
    if (a < 0.25f) return sgn | (uint8_t)qi;                        // subnormal: code = multiple of 2^-4 (4 -> the first normal)
    int ee = ex + 3, m = qi - 4;
    if (m == 4) { m = 0; ++ee; }
    return sgn | (uint8_t)((ee << 2) | m);
}
inline int pow2_shift_for(float amax, int top)                     // s with amax * 2^s in [2^top, 2^(top + 1))
{
    if (!(amax > 0.0f)) return 0;
    int e;
    frexpf(amax, &e);
    return top - (e - 1);
}
// Per-OUTPUT-CHANNEL shifts of the two correction operands of a filter [C][C][3][3]: out[o] for w, out[C + o] for
// w - f16(w), each the power of two that brings the row's largest magnitude into [2^top, 2^(top + 1)) (clamped to a signed
// byte; an all-zero row: 0).  ints[0], ints[1]: the smallest of each kind (the shift a single scale per tensor would be).
inline void row_shifts(const float* w_oihw, int C, int top, int8_t* out, int32_t* ints)
{
    int min_h = 127, min_l = 127;
    for (int o = 0; o < C; ++o) {
        float wmax = 0.0f, lmax = 0.0f;
        for (size_t i = 0; i < (size_t)C * 9; ++i) {
            const float w = w_oihw[(size_t)o * C * 9 + i], l = w - f16_bits_to_f32(f32_to_f16_bits(w));
            wmax = fabsf(w) > wmax ? fabsf(w) : wmax;
            lmax = fabsf(l) > lmax ? fabsf(l) : lmax;
        }
        int sh = pow2_shift_for(wmax, top), sl = pow2_shift_for(lmax, top);
        sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
        sl = sl > 100 ? 100 : (sl < -100 ? -100 : sl);
        out[o] = (int8_t)sh;
        out[C + o] = (int8_t)sl;
        if (wmax > 0.0f && sh < min_h) min_h = sh;
        if (lmax > 0.0f && sl < min_l) min_l = sl;
    }
    ints[0] = min_h == 127 ? 0 : min_h;
    ints[1] = min_l == 127 ? 0 : min_l;
}
}  // namespace

extern "C" size_t cz_conv3x3_c8_packed_bytes(int channels)
{
    if (channels != 128 && channels != 192) return 0;
    const size_t C = channels, KK = C / 16, CT = C / 32, NB = C / 64;
    const size_t main_u4 = (9 * KK + W_PAD_STEPS) * CT * 64, c8_u4 = (9 * NB + 1) * 2 * CT * 2 * 64;
    return (main_u4 + c8_u4 + 1) * 16 + 2 * C;          // fragments, 4 ints, 2 x C per-output-channel shifts (signed bytes)
}

extern "C" int cz_conv3x3_c8_pack_weights(const float* w_oihw, int channels, void* out_host)
{
    if (!w_oihw || !out_host || (channels != 128 && channels != 192)) {
        czi_set_error("cz_conv3x3_c8_pack_weights: bad argument (128 or 192 filters)");
        return CZ_ERR_ARG;
    }
    const int C = channels, KK = C / 16, CT = C / 32, NB = C / 64;
    const size_t main_u4 = (size_t)(9 * KK + W_PAD_STEPS) * CT * 64, c8_u4 = (size_t)(9 * NB + 1) * 2 * CT * 2 * 64;
    memset(out_host, 0, (main_u4 + c8_u4 + 1) * 16 + 2 * (size_t)C);
    uint16_t* hi = (uint16_t*)out_host;
    uint8_t* c8p = (uint8_t*)out_host + main_u4 * 16;
    int32_t* sc = (int32_t*)((uint8_t*)out_host + (main_u4 + c8_u4) * 16);
    int8_t* row_sh = (int8_t*)(sc + 4);                  // per output channel: shift of e4m3(w), then (row_sh + C) of e4m3(w - f16(w))
    row_shifts(w_oihw, C, 7, row_sh, sc);                // largest magnitude of a row in [128, 256): below 448
    for (int tap = 0; tap < 9; ++tap)
        for (int ct = 0; ct < CT; ++ct)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = ct * 32 + (lane & 31);
                for (int kk = 0; kk < KK; ++kk)
                    for (int j = 0; j < 8; ++j) {
                        const int c = kk * 16 + (lane >> 5) * 8 + j;
                        const float w = w_oihw[((size_t)o * C + c) * 9 + tap];
                        hi[((((size_t)tap * KK + kk) * CT + ct) * 64 + lane) * 8 + j] = f32_to_f16_bits(w);
                    }
                for (int b = 0; b < NB; ++b)
                    for (int q = 0; q < 2; ++q)
                        for (int j = 0; j < 32; ++j) {
                            const int c = b * 64 + (lane >> 5) * 32 + j;
                            const float w = w_oihw[((size_t)o * C + c) * 9 + tap];
                            const float v = q == 0 ? ldexpf(w, row_sh[o]) : ldexpf(w - f16_bits_to_f32(f32_to_f16_bits(w)), row_sh[C + o]);
                            const size_t u4 = ((((size_t)(tap * NB + b) * 2 + q) * CT + ct) * 2 + j / 16) * 64 + lane;
                            c8p[u4 * 16 + j % 16] = f32_to_e4m3_bits(v);
                        }
            }
    return CZ_OK;
}

// The same filter for the c6 arithmetic (k_resblock_c8<.., C6>): fp16 fragments as above; the correction pieces in bf6,
// 24 bytes per lane -- a 16-byte head where the e4m3 piece's first half sits, the 8-byte tails densely behind the heads of
// a (block, kind, wave) group -- with element e of lane (row, half) = input channel 64 b + 32 half + 8 (e >> 3) +
// ((e >> 1) & 3) + 4 (e & 1) (the order the kernels' conversion instruction gives the activations).  Trailing ints:
// the two filter shifts (largest magnitude in [8, 16): bf6 saturates at 28), then x_exp / y_exp: the exponents k of the
// activation image this convolution reads and of the one it writes (x_hi6 = bf6(x 2^-k); 2^k 28 >= max |x|).
extern "C" int cz_conv3x3_c6_pack_weights(const float* w_oihw, int channels, int x_exp, int y_exp, void* out_host)
{
    if (!w_oihw || !out_host || (channels != 128 && channels != 192) || x_exp < -100 || x_exp > 100 ||
        ((y_exp < -100 || y_exp > 100) && y_exp != CZ_C6_OUT_C8)) {
        czi_set_error("cz_conv3x3_c6_pack_weights: bad argument (128 or 192 filters; image exponents within +-100, or y_exp 127 = c8 output)");
        return CZ_ERR_ARG;
    }
    const int C = channels, KK = C / 16, CT = C / 32, NB = C / 64;
    const size_t main_u4 = (size_t)(9 * KK + W_PAD_STEPS) * CT * 64, c8_u4 = (size_t)(9 * NB + 1) * 2 * CT * 2 * 64;
    memset(out_host, 0, (main_u4 + c8_u4 + 1) * 16 + 2 * (size_t)C);
    uint16_t* hi = (uint16_t*)out_host;
    uint8_t* c6p = (uint8_t*)out_host + main_u4 * 16;
    int32_t* sc = (int32_t*)((uint8_t*)out_host + (main_u4 + c8_u4) * 16);
    int8_t* row_sh = (int8_t*)(sc + 4);
    row_shifts(w_oihw, C, 3, row_sh, sc);                // largest magnitude of a row in [8, 16): bf6 saturates at 28
    sc[2] = x_exp;
    sc[3] = y_exp;
    for (int tap = 0; tap < 9; ++tap)
        for (int ct = 0; ct < CT; ++ct)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = ct * 32 + (lane & 31);
                for (int kk = 0; kk < KK; ++kk)
                    for (int j = 0; j < 8; ++j) {
                        const int c = kk * 16 + (lane >> 5) * 8 + j;
                        const float w = w_oihw[((size_t)o * C + c) * 9 + tap];
                        hi[((((size_t)tap * KK + kk) * CT + ct) * 64 + lane) * 8 + j] = f32_to_f16_bits(w);
                    }
                for (int b = 0; b < NB; ++b)
                    for (int q = 0; q < 2; ++q) {
                        uint8_t piece[24] = {0};
                        for (int e = 0; e < 32; ++e) {
                            const int c = b * 64 + (lane >> 5) * 32 + 8 * (e >> 3) + ((e >> 1) & 3) + 4 * (e & 1);
                            const float w = w_oihw[((size_t)o * C + c) * 9 + tap];
                            const float v = q == 0 ? ldexpf(w, row_sh[o]) : ldexpf(w - f16_bits_to_f32(f32_to_f16_bits(w)), row_sh[C + o]);
                            const uint32_t code = f32_to_bf6_bits(v);
                            const int bit = 6 * e;
                            piece[bit >> 3] |= (uint8_t)(code << (bit & 7));
                            if ((bit & 7) > 2) piece[(bit >> 3) + 1] |= (uint8_t)(code >> (8 - (bit & 7)));
                        }
                        uint8_t* grp = c6p + ((((size_t)(tap * NB + b) * 2 + q) * CT + ct) * 2) * 64 * 16;    // 2 KB per (block, kind, wave)
                        memcpy(grp + (size_t)lane * 16, piece, 16);
                        memcpy(grp + 1024 + (size_t)lane * 8, piece + 16, 8);
                    }
            }
    return CZ_OK;
}

extern "C" int cz_conv3x3_c8(const void* x_hi, const void* x_c8, const void* w_packed, const float* bias,
                             const void* skip_hi, const void* skip_c8, void* y_hi, void* y_c8, float* y_f32,
                             int n_boards, int channels, int relu, void* stream)
{
    if (n_boards < 0 || !x_hi || !x_c8 || !w_packed || !bias || (channels != 128 && channels != 192) ||
        (!y_f32 && (!y_hi || !y_c8)) || (skip_hi && !skip_c8)) {
        czi_set_error("cz_conv3x3_c8: bad argument (128 or 192 filters; output: y_f32, or the operand pair y_hi + y_c8)");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    if (channels == 128) {
        constexpr int P = 2;
        hipLaunchKernelGGL((k_conv3x3_c8<128, P>), dim3((unsigned)((n_boards + P - 1) / P)), dim3(128 / 32 * 64), 0,
                           (hipStream_t)stream, (const _Float16*)x_hi, (const unsigned char*)x_c8, (const uint4*)w_packed, bias,
                           (const _Float16*)skip_hi, (const unsigned char*)skip_c8, (_Float16*)y_hi, (unsigned char*)y_c8,
                           y_f32, n_boards, relu);
    } else {
        hipLaunchKernelGGL((k_conv3x3_c8<192, 1>), dim3((unsigned)n_boards), dim3(192 / 32 * 64), 0,
                           (hipStream_t)stream, (const _Float16*)x_hi, (const unsigned char*)x_c8, (const uint4*)w_packed, bias,
                           (const _Float16*)skip_hi, (const unsigned char*)skip_c8, (_Float16*)y_hi, (unsigned char*)y_c8,
                           y_f32, n_boards, relu);
    }
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_conv3x3_c8: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}

extern "C" int cz_conv3x3(const void* x_hi, const void* x_lo, const void* w_packed, const float* bias,
                          const void* skip_hi, const void* skip_lo, void* y_hi, void* y_lo, float* y_f32,
                          int n_boards, int channels, int dtype, int parts, int relu, void* stream)
{
    if (n_boards < 0 || !w_packed || !bias || !x_hi || (parts == 2 && !x_lo) || (parts != 1 && parts != 2) ||
        (!y_f32 && (!y_hi || (parts == 2 && !y_lo))) || (skip_hi && parts == 2 && !skip_lo)) {
        czi_set_error("cz_conv3x3: bad argument");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == CZ_BF16)
        rc = dispatch_conv<__bf16>(channels, parts, x_hi, x_lo, w_packed, bias, skip_hi, skip_lo, y_hi, y_lo, y_f32,
                                   n_boards, relu, st);
    else if (dtype == CZ_F16)
        rc = dispatch_conv<_Float16>(channels, parts, x_hi, x_lo, w_packed, bias, skip_hi, skip_lo, y_hi, y_lo, y_f32,
                                     n_boards, relu, st);
    else
        rc = CZ_ERR_ARG;
    if (rc == CZ_ERR_ARG) czi_set_error("cz_conv3x3: unsupported channels / dtype (channels 32|128|192|256, bf16|f16)");
    else if (rc != CZ_OK) czi_set_error("cz_conv3x3: launch failed");
    return rc;
}

extern "C" size_t cz_input_conv_packed_elems(int channels, int in_planes, int parts)
{
    if (channels <= 0 || channels % 32 != 0 || parts < 1 || parts > 2 || in_planes < 1 || in_planes > 32) return 0;
    const int ic16 = (in_planes + 15) / 16;
    return (size_t)parts * (size_t)(25 + 3) * ic16 * (size_t)(channels / 32) * 64 * 8;
}

extern "C" int cz_input_conv_pack_weights(const float* w_oihw, int channels, int in_planes, int dtype, int parts,
                                          void* out_host)
{
    if (!w_oihw || !out_host || cz_input_conv_packed_elems(channels, in_planes, parts) == 0 ||
        (dtype != CZ_BF16 && dtype != CZ_F16)) {
        czi_set_error("cz_input_conv_pack_weights: bad argument");
        return CZ_ERR_ARG;
    }
    const int CT = channels / 32, ic16 = (in_planes + 15) / 16;
    const size_t part_elems = (size_t)(25 + 3) * ic16 * CT * 64 * 8;
    uint16_t* out = (uint16_t*)out_host;
    memset(out, 0, part_elems * parts * sizeof(uint16_t));
    for (int tap = 0; tap < 25; ++tap)
        for (int g = 0; g < ic16; ++g)
            for (int ct = 0; ct < CT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int o = ct * 32 + (lane & 31);
                        const int c = g * 16 + (lane >> 5) * 8 + j;
                        if (c >= in_planes) continue;
                        const float w = w_oihw[((size_t)o * in_planes + c) * 25 + tap];
                        const size_t idx = ((((size_t)tap * ic16 + g) * CT + ct) * 64 + lane) * 8 + j;
                        if (dtype == CZ_BF16) {
                            const uint16_t hi = f32_to_bf16_bits(w);
                            out[idx] = hi;
                            if (parts == 2) out[part_elems + idx] = f32_to_bf16_bits(w - bf16_bits_to_f32(hi));
                        } else {
                            const uint16_t hi = f32_to_f16_bits(w);
                            out[idx] = hi;
                            if (parts == 2) out[part_elems + idx] = f32_to_f16_bits(w - f16_bits_to_f32(hi));
                        }
                    }
    return CZ_OK;
}

namespace {
template <typename E, typename PT, int C>
int launch_input_conv(const void* planes, int in_planes, const void* wp, const float* bias, void* yh, void* yl,
                      int n, int parts, int relu, hipStream_t st)
{
    constexpr int P = 2;
    const unsigned blocks = (unsigned)((n + P - 1) / P);
#define CZ_IC(IC16, PARTS)                                                                                       \
    hipLaunchKernelGGL((k_input_conv<E, PT, C, IC16, P, PARTS>), dim3(blocks), dim3(C / 32 * 64), 0, st,           \
                       (const PT*)planes, (const E*)wp, bias, (E*)yh, (E*)yl, n, in_planes, relu, g_q.rows, g_q.n_dev)
    if (in_planes <= 16) {
        if (parts == 2) CZ_IC(1, 2); else CZ_IC(1, 1);
    } else {
        if (parts == 2) CZ_IC(2, 2); else CZ_IC(2, 1);
    }
#undef CZ_IC
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

template <typename E, typename PT>
int dispatch_input_conv(int channels, const void* planes, int in_planes, const void* wp, const float* bias, void* yh,
                        void* yl, int n, int parts, int relu, hipStream_t st)
{
    if (channels == 128) return launch_input_conv<E, PT, 128>(planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    if (channels == 192) return launch_input_conv<E, PT, 192>(planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    if (channels == 256) return launch_input_conv<E, PT, 256>(planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    if (channels == 32) return launch_input_conv<E, PT, 32>(planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    return CZ_ERR_ARG;
}

template <typename E>
int dispatch_input_conv_pt(int planes_dtype, int channels, const void* planes, int in_planes, const void* wp,
                           const float* bias, void* yh, void* yl, int n, int parts, int relu, hipStream_t st)
{
    switch (planes_dtype) {
    case CZ_F32: return dispatch_input_conv<E, float>(channels, planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    case CZ_F16: return dispatch_input_conv<E, _Float16>(channels, planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    case CZ_BF16: return dispatch_input_conv<E, __bf16>(channels, planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    case CZ_U8: return dispatch_input_conv<E, unsigned char>(channels, planes, in_planes, wp, bias, yh, yl, n, parts, relu, st);
    }
    return CZ_ERR_ARG;
}
}  // namespace

extern "C" int cz_input_conv(const void* planes, int planes_dtype, int in_planes, const void* w_packed,
                             const float* bias, void* y_hi, void* y_lo, int n_boards, int channels, int dtype,
                             int parts, int relu, void* stream)
{
    if (n_boards < 0 || !planes || !w_packed || !bias || !y_hi || (parts == 2 && !y_lo) || (parts != 1 && parts != 2) ||
        in_planes < 1 || in_planes > 32) {
        czi_set_error("cz_input_conv: bad argument");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = CZ_ERR_ARG;
    if (dtype == CZ_BF16)
        rc = dispatch_input_conv_pt<__bf16>(planes_dtype, channels, planes, in_planes, w_packed, bias, y_hi, y_lo,
                                            n_boards, parts, relu, st);
    else if (dtype == CZ_F16)
        rc = dispatch_input_conv_pt<_Float16>(planes_dtype, channels, planes, in_planes, w_packed, bias, y_hi, y_lo,
                                              n_boards, parts, relu, st);
    else if (dtype == CZ_F16C8 && (channels == 128 || channels == 192) && parts == 2 &&
             (planes_dtype == CZ_U8 || planes_dtype == CZ_F32)) {
        // f16-split filters (cz_input_conv_pack_weights with CZ_F16), output = the c8 operand pair (y_lo = the c8 image)
        constexpr int P = 2;
        const unsigned blocks = (unsigned)((n_boards + P - 1) / P);
#define CZ_IC8(CH, PT, IC16)                                                                                          \
        hipLaunchKernelGGL((k_input_conv<_Float16, PT, CH, IC16, P, 2, true>), dim3(blocks), dim3(CH / 32 * 64), 0, st,   \
                           (const PT*)planes, (const _Float16*)w_packed, bias, (_Float16*)y_hi, (_Float16*)y_lo, n_boards,    \
                           in_planes, relu, g_q.rows, g_q.n_dev)
#define CZ_IC8C(CH)                                                                                                   \
        do {                                                                                                          \
            if (planes_dtype == CZ_U8) { if (in_planes <= 16) CZ_IC8(CH, unsigned char, 1); else CZ_IC8(CH, unsigned char, 2); } \
            else { if (in_planes <= 16) CZ_IC8(CH, float, 1); else CZ_IC8(CH, float, 2); }                              \
        } while (0)
        if (channels == 128) CZ_IC8C(128); else CZ_IC8C(192);
#undef CZ_IC8C
#undef CZ_IC8
        rc = hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
    }
    if (rc == CZ_ERR_ARG) czi_set_error("cz_input_conv: unsupported channels / dtype");
    else if (rc != CZ_OK) czi_set_error("cz_input_conv: launch failed");
    return rc;
}

namespace {
template <typename E, int C, int PARTS, int P, bool HEADS = false, int CTW = 1>
int launch_resblock(const void* xh, const void* xl, const void* w1, const float* b1, const void* w2, const float* b2,
                    void* yh, void* yl, float* yf, int n, int n_cu, hipStream_t st, HeadArgs hd = HeadArgs{})
{
    const int tiles = (n + P - 1) / P;
    const unsigned blocks = (unsigned)(tiles < n_cu ? tiles : n_cu);
    hipLaunchKernelGGL((k_resblock<E, C, PARTS, P, HEADS, CTW>), dim3(blocks), dim3((C / 32 / CTW + 4) * 64), 0, st,
                       (const E*)xh, (const E*)xl, (const E*)w1, b1, (const E*)w2, b2, (E*)yh, (E*)yl, yf, n, hd,
                       g_q.n_dev);
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

template <bool FIRST, bool HEADS, bool C6 = false>
int launch_resblock_c8(const void* xh, const void* xc, const void* w1, const float* b1, const void* w2, const float* b2,
                       void* yh, void* yc, float* yf, int n, int n_cu, hipStream_t st, HeadArgs hd, const int32_t* n_dev,
                       FirstArgs fa)
{
    const unsigned blocks = (unsigned)(n < n_cu ? n : n_cu);
    hipLaunchKernelGGL((k_resblock_c8<FIRST, HEADS, C6>), dim3(blocks), dim3(512), 0, st, (const _Float16*)xh,
                       (const unsigned char*)xc, w1, b1, w2, b2, (_Float16*)yh, (unsigned char*)yc, yf, n, hd, n_dev, fa);
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

int g_resblock_pipelined = 1;       // cz_resblock_pipelined(): 128-filter split blocks on k_resblock_pipe
int g_first_w1_rounds = 6;          // cz_input_resblock: gather rounds done under K loop 1 (tuning hook: CZ_FIRST_W1_ROUNDS)

template <typename E>
int dispatch_resblock(int channels, int parts, const void* xh, const void* xl, const void* w1, const float* b1,
                      const void* w2, const float* b2, void* yh, void* yl, float* yf, int n, int n_cu, hipStream_t st)
{
#define CZ_RB_ARGS xh, xl, w1, b1, w2, b2, yh, yl, yf, n, n_cu, st
    if (channels == 128 && parts == 2 && !yf && g_resblock_pipelined) {
        // (operand-pair output: the software-pipelined kernel; the last block of a tower -- fp32 / head output --
        //  stays on k_resblock)
        const unsigned blocks = (unsigned)(n < n_cu ? n : n_cu);
        hipLaunchKernelGGL((k_resblock_pipe<E>), dim3(blocks), dim3(512), 0, st, (const E*)xh, (const E*)xl,
                           (const E*)w1, b1, (const E*)w2, b2, (E*)yh, (E*)yl, n, g_q.n_dev, FirstArgs{});
        return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
    }
    if (channels == 128 && parts == 2) {
        return launch_resblock<E, 128, 2, 1>(CZ_RB_ARGS);
    }
    if (channels == 192 && parts == 2) {
        const unsigned blocks = (unsigned)(n < n_cu ? n : n_cu);
        hipLaunchKernelGGL((k_resblock_ip<E, 192>), dim3(blocks), dim3(192 / 32 * 64 + ip::COPY_THREADS), 0, st,
                           (const E*)xh, (const E*)xl, (const E*)w1, b1, (const E*)w2, b2, (E*)yh, (E*)yl, yf, n, g_q.n_dev);
        return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
    }
    if (channels == 128 && parts == 1) return launch_resblock<E, 128, 1, 2>(CZ_RB_ARGS);
    if (channels == 192 && parts == 1) return launch_resblock<E, 192, 1, 1>(CZ_RB_ARGS);
    // 256 filters, plain operands: two channel tiles per matrix wave (4 + 4 waves), see conv_kloop; the tuning hook's
    // 0 keeps the one-tile schedule (8 + 4 waves) for A/B runs
    if (channels == 256 && parts == 1 && g_resblock_pipelined) return launch_resblock<E, 256, 1, 1, false, 2>(CZ_RB_ARGS);
    if (channels == 256 && parts == 1) return launch_resblock<E, 256, 1, 1>(CZ_RB_ARGS);
#undef CZ_RB_ARGS
    return CZ_ERR_ARG;
}
}  // namespace

static int device_cu_count()
{
    static int n_cu = 0;
    if (n_cu == 0) {
        if (const char* e = getenv("CZ_FIRST_W1_ROUNDS")) g_first_w1_rounds = atoi(e);      // (A/B runs of the fused input layer)
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        n_cu = prop.multiProcessorCount;
    }
    return n_cu;
}

extern "C" int cz_resblock_heads(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1,
                                 const void* w2_packed, const float* bias2, const float* head_w, const float* head_b,
                                 float* policy_feat, float* value_feat, int n_boards, int channels, int dtype,
                                 int n_policy, int n_value, void* stream)
{
    if (n_boards < 0 || !x_hi || !x_lo || !w1_packed || !w2_packed || !bias1 || !bias2 || !head_w || !head_b ||
        !policy_feat || !value_feat || n_policy < 1 || n_value < 1 || n_policy + n_value != 6) {
        czi_set_error("cz_resblock_heads: bad argument (split operands; n_policy + n_value == 6)");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    if (channels != 128 || (dtype != CZ_BF16 && dtype != CZ_F16 && dtype != CZ_F16C8 && dtype != CZ_F16C6)) {
        czi_set_error("cz_resblock_heads: 128 filters, bf16 / f16 split operands only (use cz_resblock + cz_head_convs)");
        return CZ_ERR_ARG;
    }
    const int n_cu = device_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_resblock_heads: cannot query the device");
        return CZ_ERR_HIP;
    }
    const HeadArgs hd{head_w, head_b, policy_feat, value_feat, n_policy};
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == CZ_BF16)
        rc = launch_resblock<__bf16, 128, 2, 1, true>(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, nullptr, nullptr,
                                                       nullptr, n_boards, n_cu, st, hd);
    else if (dtype == CZ_F16C8)
        rc = launch_resblock_c8<false, true>(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, nullptr, nullptr, nullptr,
                                             n_boards, n_cu, st, hd, g_q.n_dev, FirstArgs{});
    else if (dtype == CZ_F16C6)
        rc = launch_resblock_c8<false, true, true>(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, nullptr, nullptr, nullptr,
                                                   n_boards, n_cu, st, hd, g_q.n_dev, FirstArgs{});
    else
        rc = launch_resblock<_Float16, 128, 2, 1, true>(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, nullptr,
                                                         nullptr, nullptr, n_boards, n_cu, st, hd);
    if (rc != CZ_OK) czi_set_error("cz_resblock_heads: launch failed");
    return rc;
}

extern "C" int cz_resblock(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1,
                           const void* w2_packed, const float* bias2, void* y_hi, void* y_lo, float* y_f32,
                           int n_boards, int channels, int dtype, int parts, void* stream)
{
    if (n_boards < 0 || !x_hi || !w1_packed || !w2_packed || !bias1 || !bias2 || (parts != 1 && parts != 2) ||
        (parts == 2 && (!x_lo || (!y_f32 && (!y_hi || !y_lo)))) || (parts == 1 && (!y_hi || y_f32))) {
        czi_set_error("cz_resblock: bad argument (parts = 1 writes y_hi only; y_f32 needs parts = 2)");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = device_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_resblock: cannot query the device");
        return CZ_ERR_HIP;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = CZ_ERR_ARG;
    if (dtype == CZ_BF16)
        rc = dispatch_resblock<__bf16>(channels, parts, x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo,
                                       y_f32, n_boards, n_cu, st);
    else if (dtype == CZ_F16)
        rc = dispatch_resblock<_Float16>(channels, parts, x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo,
                                         y_f32, n_boards, n_cu, st);
    else if (dtype == CZ_F16C8 && channels == 128 && parts == 2)
        rc = launch_resblock_c8<false, false>(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo, y_f32, n_boards,
                                              n_cu, st, HeadArgs{}, g_q.n_dev, FirstArgs{});
    else if (dtype == CZ_F16C6 && channels == 128 && parts == 2)
        rc = launch_resblock_c8<false, false, true>(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo, y_f32, n_boards,
                                                    n_cu, st, HeadArgs{}, g_q.n_dev, FirstArgs{});
    else if ((dtype == CZ_F16C8 || dtype == CZ_F16C6 || dtype == CZ_F16C86) && channels == 192 && parts == 2) {
        // 192 filters: the two-image in-place block on c8, on c6 (round 6), or as the tower's first c6 block behind the input
        // layer's c8 image (CZ_F16C86: the first filter is cz_conv3x3_c8_pack_weights', the second cz_conv3x3_c6_pack_weights')
        ip::Chain ch{};
        ch.n = 1;
        ch.w1[0] = w1_packed; ch.w2[0] = w2_packed; ch.b1[0] = bias1; ch.b2[0] = bias2;
        // CZ_IP_PAIR=0: one board on six matrix waves (k_resblock_ip_c8); default: a pair on four waves of three channel tiles
        const char* pair_env = getenv("CZ_IP_PAIR");
        const bool pair = !(pair_env && pair_env[0] == '0');
        const int units = pair ? (n_boards + 1) / 2 : n_boards;
        const unsigned blocks = (unsigned)(units < n_cu ? units : n_cu);
#define CZ_IP_LAUNCH(XF, YF) do { \
            if (pair) hipLaunchKernelGGL((k_resblock_ip4_c8<192, XF, YF>), dim3(blocks), dim3(256), 0, st, (const _Float16*)x_hi, \
                                         (const unsigned char*)x_lo, ch, (_Float16*)y_hi, (unsigned char*)y_lo, y_f32, n_boards, g_q.n_dev, 0, HeadArgs{}); \
            else hipLaunchKernelGGL((k_resblock_ip_c8<192, XF, YF>), dim3(blocks), dim3(192 / 32 * 64 + ip::COPY_THREADS), 0, st, \
                                    (const _Float16*)x_hi, (const unsigned char*)x_lo, ch, (_Float16*)y_hi, (unsigned char*)y_lo, y_f32, \
                                    n_boards, g_q.n_dev); } while (0)
        if (dtype == CZ_F16C8) CZ_IP_LAUNCH(0, 0);
        else if (dtype == CZ_F16C6) CZ_IP_LAUNCH(1, 1);
        else CZ_IP_LAUNCH(0, 1);
#undef CZ_IP_LAUNCH
        rc = hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
    }
    if (rc == CZ_ERR_ARG)
        czi_set_error("cz_resblock: supported: 128 / 192 filters (split or plain operands), 256 filters (plain), bf16 / f16; "
                      "use cz_conv3x3 otherwise");
    else if (rc != CZ_OK)
        czi_set_error("cz_resblock: launch failed");
    return rc;
}

// n_blocks (1 .. 12) consecutive residual blocks of a 192-filter tower on ONE staged arithmetic in one launch (k_resblock_ip_c8's
// chain): dtype CZ_F16C8 (c8 blocks) or CZ_F16C6 (c6 blocks behind the tower's first one).  y_f32 != NULL: the last block writes
// fp32 instead of the operand pair.
extern "C" int cz_resblock_chain(const void* x_hi, const void* x_img, int n_blocks, const void* const* w1_packed,
                                 const float* const* bias1, const void* const* w2_packed, const float* const* bias2, void* y_hi,
                                 void* y_img, float* y_f32, int n_boards, int channels, int dtype, const int32_t* n_dev,
                                 void* stream)
{
    const bool pairs = dtype == CZ_F16 || dtype == CZ_BF16;     // (hi, lo) pair blocks: x_img / y_img are the lo tensors
    if (n_boards < 0 || !x_hi || !x_img || !w1_packed || !w2_packed || !bias1 || !bias2 || n_blocks < 1 ||
        n_blocks > ip::MAX_BLOCKS || channels != 192 || (dtype != CZ_F16C8 && dtype != CZ_F16C6 && dtype != CZ_F16C86 && !pairs) ||
        (!y_f32 && (!y_hi || !y_img))) {
        czi_set_error("cz_resblock_chain: bad argument (192 filters, 1 .. 12 blocks, dtype CZ_F16C8 / CZ_F16C6 / CZ_F16C86 with y_f32 or y_hi + y_img, "
                      "or CZ_F16 / CZ_BF16 pair blocks with y_f32 or y_hi + y_lo)");
        return CZ_ERR_ARG;
    }
    if (pairs) {
        for (int b = 0; b < n_blocks; ++b)
            if (!w1_packed[b] || !w2_packed[b] || !bias1[b] || !bias2[b]) {
                czi_set_error("cz_resblock_chain: null block parameter");
                return CZ_ERR_ARG;
            }
        if (n_boards == 0) return CZ_OK;
        const int n_cu_p = device_cu_count();
        if (n_cu_p < 0) {
            czi_set_error("cz_resblock_chain: cannot query the device");
            return CZ_ERR_HIP;
        }
        const int rc = czi_pairs4_launch(x_hi, x_img, n_blocks, w1_packed, bias1, w2_packed, bias2, y_hi, y_img, nullptr, nullptr, nullptr,
                                         nullptr, 0, n_boards, 192, dtype, n_cu_p, n_dev, stream, y_f32);
        if (rc != CZ_OK) czi_set_error("cz_resblock_chain: launch failed");
        return rc;
    }
    ip::Chain ch{};
    ch.n = n_blocks;
    for (int b = 0; b < n_blocks; ++b) {
        if (!w1_packed[b] || !w2_packed[b] || !bias1[b] || !bias2[b]) {
            czi_set_error("cz_resblock_chain: null block parameter");
            return CZ_ERR_ARG;
        }
        ch.w1[b] = w1_packed[b]; ch.w2[b] = w2_packed[b]; ch.b1[b] = bias1[b]; ch.b2[b] = bias2[b];
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = device_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_resblock_chain: cannot query the device");
        return CZ_ERR_HIP;
    }
    hipStream_t st = (hipStream_t)stream;
    // a pair of boards per workgroup on four matrix waves of three channel tiles (k_resblock_ip4_c8);
    // CZ_IP_PAIR=0: one board on six matrix waves (k_resblock_ip_c8; A/B runs, the tests run both)
    const char* pair_env = getenv("CZ_IP_PAIR");
    if (dtype == CZ_F16C86 && pair_env && pair_env[0] == '0') {
        czi_set_error("cz_resblock_chain: CZ_F16C86 (a c6 chain that starts the tower) exists on the four-wave kernel only (CZ_IP_PAIR=0 is set)");
        return CZ_ERR_ARG;
    } else if (!(pair_env && pair_env[0] == '0')) {
        const int n_pairs = (n_boards + 1) / 2;
        const unsigned blocks = (unsigned)(n_pairs < n_cu ? n_pairs : n_cu);
        if (dtype == CZ_F16C86)
            hipLaunchKernelGGL((k_resblock_ip4_c8<192, 1, 1, true>), dim3(blocks), dim3(256), 0, st, (const _Float16*)x_hi,
                               (const unsigned char*)x_img, ch, (_Float16*)y_hi, (unsigned char*)y_img, y_f32, n_boards, n_dev, 0, HeadArgs{});
        else if (dtype == CZ_F16C8)
            hipLaunchKernelGGL((k_resblock_ip4_c8<192, 0, 0>), dim3(blocks), dim3(256), 0, st, (const _Float16*)x_hi,
                               (const unsigned char*)x_img, ch, (_Float16*)y_hi, (unsigned char*)y_img, y_f32, n_boards, n_dev, 0, HeadArgs{});
        else
            hipLaunchKernelGGL((k_resblock_ip4_c8<192, 1, 1>), dim3(blocks), dim3(256), 0, st, (const _Float16*)x_hi,
                               (const unsigned char*)x_img, ch, (_Float16*)y_hi, (unsigned char*)y_img, y_f32, n_boards, n_dev, 0, HeadArgs{});
    } else {
        const unsigned blocks = (unsigned)(n_boards < n_cu ? n_boards : n_cu);
        if (dtype == CZ_F16C8)
            hipLaunchKernelGGL((k_resblock_ip_c8<192, 0, 0>), dim3(blocks), dim3(192 / 32 * 64 + ip::COPY_THREADS), 0, st, (const _Float16*)x_hi,
                               (const unsigned char*)x_img, ch, (_Float16*)y_hi, (unsigned char*)y_img, y_f32, n_boards, n_dev);
        else
            hipLaunchKernelGGL((k_resblock_ip_c8<192, 1, 1>), dim3(blocks), dim3(192 / 32 * 64 + ip::COPY_THREADS), 0, st, (const _Float16*)x_hi,
                               (const unsigned char*)x_img, ch, (_Float16*)y_hi, (unsigned char*)y_img, y_f32, n_boards, n_dev);
    }
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_resblock_chain: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}

// cz_tower's launches on the four-wave kernel (csrc/xq_tower.hip): a chain of 128-filter blocks of one staged arithmetic
// (c6: 1, c8: 0); exit: 0 = the operand pair (c6 image, or the c8 image a c6 chain hands over), IP4_EXIT_PAIRS, IP4_EXIT_HEADS.
extern "C" int czi_tower4_launch(const void* x_hi, const void* x_img, int n_blocks, const void* const* w1, const float* const* b1,
                                 const void* const* w2, const float* const* b2, int c6, int exit_mode, void* y_hi, void* y_img,
                                 const float* head_w, const float* head_b, float* pol, float* val, int n_pol, int n_boards,
                                 int n_cu, const int32_t* n_dev, void* stream)
{
    if (n_blocks > ip::MAX_BLOCKS) return CZ_ERR_ARG;
    const HeadArgs hd{head_w, head_b, pol, val, n_pol};
    hipStream_t st = (hipStream_t)stream;
    ip::Chain ch{};
    ch.n = n_blocks;
    for (int b = 0; b < n_blocks; ++b) { ch.w1[b] = w1[b]; ch.w2[b] = w2[b]; ch.b1[b] = b1[b]; ch.b2[b] = b2[b]; }
    const int n_pairs = (n_boards + 1) / 2;
    const unsigned blocks = (unsigned)(n_pairs < n_cu ? n_pairs : n_cu);
    if (c6)
        hipLaunchKernelGGL((k_resblock_ip4_c8<128, 1, 1>), dim3(blocks), dim3(256), 0, st, (const _Float16*)x_hi, (const unsigned char*)x_img,
                           ch, (_Float16*)y_hi, (unsigned char*)y_img, (float*)nullptr, n_boards, n_dev, exit_mode, hd);
    else
        hipLaunchKernelGGL((k_resblock_ip4_c8<128, 0, 0>), dim3(blocks), dim3(256), 0, st, (const _Float16*)x_hi, (const unsigned char*)x_img,
                           ch, (_Float16*)y_hi, (unsigned char*)y_img, (float*)nullptr, n_boards, n_dev, exit_mode, hd);
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

// n_blocks (1 .. 24) consecutive residual blocks of a 256-filter tower on plain fp16 / bf16 operands in one launch
// (k_tower_plain2: a pair of boards per workgroup, one LDS image per board; k_tower_plain with CZ_TOWER_PLAIN_PAIR=0): the `deep`
// 20 x 256 configuration is ONE launch.  Filters: cz_conv3x3_pack_weights(parts = 1).
extern "C" int cz_tower_plain(const void* x, int n_blocks, const void* const* w1_packed, const float* const* bias1,
                              const void* const* w2_packed, const float* const* bias2, void* y, int n_boards, int channels,
                              int dtype, const int32_t* n_dev, void* stream)
{
    if (n_boards < 0 || !x || !y || !w1_packed || !w2_packed || !bias1 || !bias2 || n_blocks < 1 || n_blocks > pl::MAX_BLOCKS ||
        channels != 256 || (dtype != CZ_F16 && dtype != CZ_BF16)) {
        czi_set_error("cz_tower_plain: bad argument (256 filters, plain f16 / bf16 operands, 1 .. 24 blocks)");
        return CZ_ERR_ARG;
    }
    pl::Chain ch{};
    ch.n = n_blocks;
    for (int b = 0; b < n_blocks; ++b) {
        if (!w1_packed[b] || !w2_packed[b] || !bias1[b] || !bias2[b]) {
            czi_set_error("cz_tower_plain: null block parameter");
            return CZ_ERR_ARG;
        }
        ch.w1[b] = w1_packed[b]; ch.w2[b] = w2_packed[b]; ch.b1[b] = bias1[b]; ch.b2[b] = bias2[b];
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = device_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_tower_plain: cannot query the device");
        return CZ_ERR_HIP;
    }
    hipStream_t st = (hipStream_t)stream;
    // CZ_TOWER_PLAIN_PAIR=0: one board per workgroup in X | Y (k_tower_plain, round 6's first version; A/B and the tests)
    const char* pair_env = getenv("CZ_TOWER_PLAIN_PAIR");       // (read per call: the tests run both)
    const bool pair = !(pair_env && pair_env[0] == '0');
    if (pair) {
        const int n_pairs = (n_boards + 1) / 2;
        const unsigned blocks = (unsigned)(n_pairs < n_cu ? n_pairs : n_cu);
        if (dtype == CZ_F16)
            hipLaunchKernelGGL((k_tower_plain2<_Float16, 256, 2>), dim3(blocks), dim3(256 / 32 / 2 * 64), 0, st, (const _Float16*)x, ch,
                               (_Float16*)y, n_boards, n_dev);
        else
            hipLaunchKernelGGL((k_tower_plain2<__bf16, 256, 2>), dim3(blocks), dim3(256 / 32 / 2 * 64), 0, st, (const __bf16*)x, ch,
                               (__bf16*)y, n_boards, n_dev);
    } else {
        const unsigned blocks = (unsigned)(n_boards < n_cu ? n_boards : n_cu);
        if (dtype == CZ_F16)
            hipLaunchKernelGGL((k_tower_plain<_Float16, 256, 2>), dim3(blocks), dim3((256 / 32 / 2 + 4) * 64), 0, st, (const _Float16*)x, ch,
                               (_Float16*)y, n_boards, n_dev);
        else
            hipLaunchKernelGGL((k_tower_plain<__bf16, 256, 2>), dim3(blocks), dim3((256 / 32 / 2 + 4) * 64), 0, st, (const __bf16*)x, ch,
                               (__bf16*)y, n_boards, n_dev);
    }
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_tower_plain: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}

// The input layer and the first residual block in one launch (k_resblock_pipe<FIRST>): the 5 x 5 input convolution of
// the one-hot feature planes is a gather over the occupied squares, done by the block's copy waves.
extern "C" int cz_input_resblock_m(const void* planes_u8, const uint32_t* masks, int in_planes, const float* in_table,
                                   const float* in_bias, const void* w1_packed, const float* bias1, const void* w2_packed,
                                   const float* bias2, void* y_hi, void* y_lo, int n_boards, int channels, int dtype,
                                   const int32_t* rows, const int32_t* n_dev, void* stream);

extern "C" int cz_input_resblock(const void* planes_u8, int in_planes, const float* in_table, const float* in_bias,
                                 const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2,
                                 void* y_hi, void* y_lo, int n_boards, int channels, int dtype, const int32_t* rows,
                                 const int32_t* n_dev, void* stream)
{
    return cz_input_resblock_m(planes_u8, nullptr, in_planes, in_table, in_bias, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo,
                               n_boards, channels, dtype, rows, n_dev, stream);
}

// ... with the positions' occupancy boards handed in (masks [n][96] uint32 DEVICE, word = plane position, bit c = plane c shows
// a piece there: what cz_search_leaf_masks makes the search kernel write beside the planes): the copy waves skip deriving them
// from the 1260 plane bytes.  masks = NULL: exactly cz_input_resblock.  With masks the planes are not read at all.
extern "C" int cz_input_resblock_m(const void* planes_u8, const uint32_t* masks, int in_planes, const float* in_table,
                                   const float* in_bias, const void* w1_packed, const float* bias1, const void* w2_packed,
                                   const float* bias2, void* y_hi, void* y_lo, int n_boards, int channels, int dtype,
                                   const int32_t* rows, const int32_t* n_dev, void* stream)
{
    if (n_boards < 0 || (!planes_u8 && !masks) || !in_table || !in_bias || !w1_packed || !w2_packed || !bias1 || !bias2 || !y_hi ||
        !y_lo || in_planes < 1 || in_planes > 32 || (in_planes * 90) % 4 != 0) {
        czi_set_error("cz_input_resblock: bad argument (u8 planes, in_planes even and <= 32)");
        return CZ_ERR_ARG;
    }
    if (channels != 128 || (dtype != CZ_BF16 && dtype != CZ_F16 && dtype != CZ_F16C8 && dtype != CZ_F16C6)) {
        czi_set_error("cz_input_resblock: 128 filters; bf16 / f16 split operands or the c8 / c6 pair (use cz_input_conv + cz_resblock)");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = device_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_input_resblock: cannot query the device");
        return CZ_ERR_HIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(n_boards < n_cu ? n_boards : n_cu);
    // 6 of the ~12-16 term rounds of a board under K loop 1, the rest under K loop 2: measured on one box, extra time of
    // the launch against an inner block's: 0 rounds +0.43 ms, 3: +0.26, 6: +0.14, 9: +0.22 (window 2 also drains the result)
    const FirstArgs fa{(const unsigned char*)planes_u8, in_table, in_bias, rows, masks, in_planes, g_first_w1_rounds};
    if (dtype == CZ_F16C6) {          // y_lo = a c6 image; w1: cz_conv3x3_c8_pack_weights' (the gather's image is c8), w2: ..._c6_...
        if (launch_resblock_c8<true, false, true>(nullptr, nullptr, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo, nullptr,
                                                  n_boards, n_cu, st, HeadArgs{}, n_dev, fa) != CZ_OK) {
            czi_set_error("cz_input_resblock: launch failed");
            return CZ_ERR_HIP;
        }
        return CZ_OK;
    }
    if (dtype == CZ_F16C8) {          // y_lo = the c8 image, the filters are cz_conv3x3_c8_pack_weights' (k_resblock_c8<FIRST>)
        if (launch_resblock_c8<true, false>(nullptr, nullptr, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo, nullptr, n_boards,
                                            n_cu, st, HeadArgs{}, n_dev, fa) != CZ_OK) {
            czi_set_error("cz_input_resblock: launch failed");
            return CZ_ERR_HIP;
        }
        return CZ_OK;
    }
    if (dtype == CZ_BF16)
        hipLaunchKernelGGL((k_resblock_pipe<__bf16, true>), dim3(blocks), dim3(512), 0, st, (const __bf16*)nullptr,
                           (const __bf16*)nullptr, (const __bf16*)w1_packed, bias1, (const __bf16*)w2_packed, bias2,
                           (__bf16*)y_hi, (__bf16*)y_lo, n_boards, n_dev, fa);
    else
        hipLaunchKernelGGL((k_resblock_pipe<_Float16, true>), dim3(blocks), dim3(512), 0, st, (const _Float16*)nullptr,
                           (const _Float16*)nullptr, (const _Float16*)w1_packed, bias1, (const _Float16*)w2_packed, bias2,
                           (_Float16*)y_hi, (_Float16*)y_lo, n_boards, n_dev, fa);
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_input_resblock: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}

// test / tuning hook: 1 (default) = 128-filter split residual blocks with operand-pair output run on the
// software-pipelined kernel (k_resblock_pipe), 0 = on k_resblock.  Returns the previous setting.
extern "C" int cz_resblock_pipelined(int enable)
{
    const int old = g_resblock_pipelined;
    if (enable >= 0) g_resblock_pipelined = enable ? 1 : 0;
    return old;
}

// ---- compact evaluation queue: the same kernels with a device-side board count (and a row gather in the input layer) ----
extern "C" int cz_input_conv_q(const void* planes, int planes_dtype, int in_planes, const void* w_packed,
                               const float* bias, void* y_hi, void* y_lo, int n_boards, int channels, int dtype,
                               int parts, int relu, const int32_t* rows, const int32_t* n_dev, void* stream)
{
    g_q = QueueCtx{rows, n_dev};
    const int rc = cz_input_conv(planes, planes_dtype, in_planes, w_packed, bias, y_hi, y_lo, n_boards, channels, dtype,
                                 parts, relu, stream);
    g_q = QueueCtx{};
    return rc;
}

extern "C" int cz_resblock_q(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1,
                             const void* w2_packed, const float* bias2, void* y_hi, void* y_lo, float* y_f32,
                             int n_boards, int channels, int dtype, int parts, const int32_t* n_dev, void* stream)
{
    g_q = QueueCtx{nullptr, n_dev};
    const int rc = cz_resblock(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo, y_f32, n_boards, channels,
                               dtype, parts, stream);
    g_q = QueueCtx{};
    return rc;
}

extern "C" int cz_resblock_heads_q(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1,
                                   const void* w2_packed, const float* bias2, const float* head_w, const float* head_b,
                                   float* policy_feat, float* value_feat, int n_boards, int channels, int dtype,
                                   int n_policy, int n_value, const int32_t* n_dev, void* stream)
{
    g_q = QueueCtx{nullptr, n_dev};
    const int rc = cz_resblock_heads(x_hi, x_lo, w1_packed, bias1, w2_packed, bias2, head_w, head_b, policy_feat,
                                     value_feat, n_boards, channels, dtype, n_policy, n_value, stream);
    g_q = QueueCtx{};
    return rc;
}

extern "C" int cz_split_bias_act(const float* x, const float* bias, void* y_hi, void* y_lo, size_t n_elems,
                                 int channels, int dtype, int parts, int relu, void* stream)
{
    if (!x || !y_hi || (parts == 2 && !y_lo) || (parts != 1 && parts != 2) || channels <= 0 || channels % 4 != 0 ||
        n_elems % (size_t)channels != 0 || (dtype != CZ_BF16 && dtype != CZ_F16)) {
        czi_set_error("cz_split_bias_act: bad argument");
        return CZ_ERR_ARG;
    }
    if (n_elems == 0) return CZ_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t nquad = n_elems / 4;
    size_t blocks = (nquad + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const int cquad = channels / 4;
#define CZ_SPLIT(E, PARTS)                                                                                     \
    hipLaunchKernelGGL((k_split_bias_act<E, PARTS>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias, (E*)y_hi, \
                       (E*)y_lo, nquad, cquad, relu)
    if (dtype == CZ_BF16) {
        if (parts == 2) CZ_SPLIT(__bf16, 2); else CZ_SPLIT(__bf16, 1);
    } else {
        if (parts == 2) CZ_SPLIT(_Float16, 2); else CZ_SPLIT(_Float16, 1);
    }
#undef CZ_SPLIT
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_split_bias_act: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}
