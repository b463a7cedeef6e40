// xq_nn_common.h -- device helpers shared by the network translation units (csrc/xq_conv.hip, csrc/xq_tower.hip):
// MFMA wrappers, the operand formats' conversions (c8 triple, c6 pieces), the head / input-layer argument blocks and the
// f16x3 / bf16x3 K loop with its in-place second epilogue (k_resblock_pipe's).  Everything sits in an anonymous namespace:
// each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "xq_c8_kloop.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <typename E> struct Mfma;
template <> struct Mfma<__bf16> {
    typedef bf16x8 V8;
    static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<_Float16> {
    typedef f16x8 V8;
    static __device__ __forceinline__ f32x16 mma(V8 a, V8 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int W_RING_MAX = 4;    // weight fragments in flight (register ring); 3 K-steps of prefetch
constexpr int W_PAD_STEPS = 3;   // zero K-steps appended to the packed filter so the prefetch never reads past it

template <typename E> struct alignas(8) Quad { E e[4]; };

// sum over the 16 lanes of a DPP row, on every lane of the row: four rotate-and-add steps in the VALU (round 5; the head
// convolutions' 16-lane butterflies were ds_bpermute round trips through the LDS the matrix waves read their operands from)
__device__ __forceinline__ float row16_sum(float a)
{
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x128, 0xF, 0xF, false));   // row_ror:8
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x124, 0xF, 0xF, false));   // row_ror:4
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x122, 0xF, 0xF, false));   // row_ror:2
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x121, 0xF, 0xF, false));   // row_ror:1
    return a;
}


typedef __attribute__((ext_vector_type(8))) int i32x8;
namespace cf8 {
constexpr int X_LO_SHIFT = 11;        // x_lo8 = e4m3(x_lo * 2^11): |x_lo| <= 2^-12 |x|, so |x| up to 2^9 stays below e4m3's 448
constexpr float X_LO_SCALE = 2048.0f, X_LO_INV = 1.0f / 2048.0f;
__device__ __forceinline__ float sat(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }   // e4m3's range
// four fp32 values -> the operand triple: f16(v), e4m3((v - f16(v)) * 2^11), e4m3(v)   (round to nearest even, saturating)
struct Split4 {
    Quad<_Float16> hi;
    uint32_t l8, h8;
};
__device__ __forceinline__ Split4 split4(const float* v)
{
    Split4 s;
    float lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s.hi.e[i] = (_Float16)v[i];
        lo[i] = sat((v[i] - (float)s.hi.e[i]) * X_LO_SCALE);
    }
    int l = 0, h = 0;
    l = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l, false);
    l = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l, true);
    h = __builtin_amdgcn_cvt_pk_fp8_f32(sat(v[0]), sat(v[1]), h, false);
    h = __builtin_amdgcn_cvt_pk_fp8_f32(sat(v[2]), sat(v[3]), h, true);
    s.l8 = (uint32_t)l;
    s.h8 = (uint32_t)h;
    return s;
}
// the value an operand pair stands for (skip connection): f16 + lo8 * 2^-11
__device__ __forceinline__ void add_pair4(float* v, Quad<_Float16> hi, uint32_t l8)
{
    v[0] += (float)hi.e[0] + __builtin_amdgcn_cvt_f32_fp8((int)l8, 0) * X_LO_INV;
    v[1] += (float)hi.e[1] + __builtin_amdgcn_cvt_f32_fp8((int)l8, 1) * X_LO_INV;
    v[2] += (float)hi.e[2] + __builtin_amdgcn_cvt_f32_fp8((int)l8, 2) * X_LO_INV;
    v[3] += (float)hi.e[3] + __builtin_amdgcn_cvt_f32_fp8((int)l8, 3) * X_LO_INV;
}
// where the c8 weight fragments and the two scale exponents sit behind the f16 fragments of a packed filter
template <int C> struct Pack {
    static constexpr size_t MAIN_U4 = (size_t)(9 * (C / 16) + W_PAD_STEPS) * (C / 32) * 64;
    static constexpr size_t C8_U4 = (size_t)(9 * (C / 64) + 1) * 2 * (C / 32) * 2 * 64;
};
}  // namespace cf8

// HEADS: the block is the last one of the tower and the copy waves, instead of storing its output, apply the two
// 1x1 head convolutions (6 filters: n_pol policy + 6 - n_pol value, BatchNorm folded, ReLU) to the staged fp32
// activation and write only the 6 x 90 head features per board (channels-first Flatten order).
struct HeadArgs {
    const float* w;        // [6][C]
    const float* b;        // [6]
    float* pol;            // [n][n_pol * 90]
    float* val;            // [n][(6 - n_pol) * 90]
    int n_pol;
};

struct FirstArgs {
    const unsigned char* planes;   // u8 [n][in_planes][90], 0 / 1
    const float* table;            // [in_planes][25][128]
    const float* in_bias;          // [128]
    const int32_t* rows;           // compact queue: board i of the batch is planes[rows[i]] (NULL: identity)
    const uint32_t* masks;         // [n][96] occupancy boards of the same positions (cz_search_leaf_masks), or NULL: then the copy
                                   // waves derive them from the planes
    int in_planes;
    int w1_rounds;                 // term rounds done in the first window (under K loop 1), the rest under K loop 2
};

constexpr int CZ_C6_OUT_C8 = 127;  // y_exp of a c6-packed filter: "the image this convolution's block writes is a c8 image"
namespace rb8 {
constexpr int RB = 256;
// ---- the c6 arithmetic (round 4): correction operands in bf6 (e3m2), 32 channels of a pixel = one 24-byte piece --------
// A piece holds the channels of one 32-block in the order v_cvt_scalef32_2xpk16_bf6_f32 packs two 16-vectors (element
// 2 i = src0[i], 2 i + 1 = src1[i], element e at bits 6 e; tools/probes/bf6_probe.hip) with src0 = the block's even
// channel quads, src1 = its odd ones -- which is what a matrix wave holds after one v_permlane32_swap per register:
//     channel of element e = 8 (e >> 3) + ((e >> 1) & 3) + 4 (e & 1)          (cz_conv3x3_c6_pack_weights: the same)
// and the inverse conversion (sequential) hands accumulator register r of lane half kb its channel at element 2 r + kb.
// In a 256-byte image row (HBM and LDS alike) piece (kind, block b, half kb) has its 16-byte head in logical chunk
// 8 kind + 4 b + 2 kb -- where the e4m3 piece's first half sits -- and its 8-byte tail at the start of the next chunk
// (c6_tail_half: with the CZ_C6_TAIL_SWZ build switch, its upper half for rows with bit 4 set -- measured and left off).
// x_hi6 = bf6(x 2^-k), x_lo6 = bf6((x - f16(x)) 2^(11 - k)) with the image's exponent k from the calibration
// (2^k * 28 >= the tensor's largest value; the conversion saturates), carried by the packed filters that read / write it.
typedef __attribute__((ext_vector_type(6))) unsigned int u32x6;
typedef __attribute__((ext_vector_type(32))) float f32x32;
__device__ __forceinline__ const int* pack_ints(const void* packed)
{
    return reinterpret_cast<const int*>(reinterpret_cast<const uint4*>(packed) + c8k::MAIN_U4 + c8k::C8_U4);
}
__device__ __forceinline__ int c6_chunk(int kind, int blk32) { return 8 * kind + 4 * (blk32 >> 1) + 2 * (blk32 & 1); }
// byte offsets of a piece's head inside a part's pixel row `row` (LDS: chunks swizzled by the row)
__device__ __forceinline__ int c6_lds_off(int row, int chunk) { return row * RB + ((chunk ^ (row & 15)) << 4); }
// byte offset of a piece's 8-byte tail inside its chunk (xq_c8_kloop.h CZ_C6_TAIL_SWZ): the upper half for rows with bit 4 set
__device__ __forceinline__ int c6_tail_half(int row) { return CZ_C6_TAIL_SWZ ? (row >> 1) & 8 : 0; }

}  // namespace rb8

namespace pipe {
constexpr int RB = 256, IMG_ROWS = 90, ROW_Z = 272, PSTR = (ROW_Z + 16) * RB;      // bytes per part
constexpr int BIAS_OFF = 2 * PSTR;                                                   // float b1[128], b2[128]
constexpr int MASK_OFF = BIAS_OFF + 2 * 128 * 4;                                     // FIRST: 4 x uint32 mask[96], one per copy wave
constexpr int LDS_BYTES = MASK_OFF + 4 * 96 * 4;
constexpr int KK = 8, NT = 3, NM = 9, W_STEP = 4 * 64, W_PART = (9 * KK + W_PAD_STEPS) * W_STEP, W_RING = 4;
}  // namespace pipe

template <typename E>
struct PipeShadow {                 // epilogue 2 of the previous board, one (tile, channel group) unit at a time
    unsigned char* lds;
    int prev_row_base;              // first absolute row of the previous board's X image
    int bias2_off = pipe::BIAS_OFF + 128 * 4;      // byte offset of the second convolution's bias vector (float[128])
    int wave, kb, ln;
    Quad<E> sh, sl;
    float4 bv;
    float v[4];
    int off;
    __device__ __forceinline__ void load(int p, int g)
    {
        const int q = p * 32 + ln;
        const int ch = wave * 32 + g * 8 + kb * 4;
        // padding pixels (q >= 90) read and write a dump in the two unused rows 270 / 271 (8 bytes per lane) instead of
        // being predicated: no exec-mask branches inside the MFMA loop
        off = q < 90 ? (prev_row_base + q) * pipe::RB + (((ch >> 3) ^ (q & 15)) << 4) + (ch & 7) * 2
                     : 270 * pipe::RB + (kb * 32 + ln) * 8;
        bv = *reinterpret_cast<const float4*>(lds + bias2_off + ch * 4);
        sh = *reinterpret_cast<const Quad<E>*>(lds + off);
        sl = *reinterpret_cast<const Quad<E>*>(lds + pipe::PSTR + off);
    }
    // the arithmetic in the order of k_resblock's epilogue 2 (bit-identical), one element and one stage at a time so
    // that no MFMA slot gets more than a few VALU instructions
    __device__ __forceinline__ void stage(const f32x16& a, int g, int st, int i)
    {
        const float b = i == 0 ? bv.x : (i == 1 ? bv.y : (i == 2 ? bv.z : bv.w));
        if (st == 0) v[i] = a[g * 4 + i] + b;
        if (st == 1) v[i] += (float)sh.e[i];
        if (st == 2) { v[i] += (float)sl.e[i]; v[i] = v[i] > 0.0f ? v[i] : 0.0f; }
        if (st == 3) { sh.e[i] = (E)v[i]; sl.e[i] = (E)(v[i] - (float)sh.e[i]); }   // (the skip registers are free again)
    }
    __device__ __forceinline__ void store()
    {
        *reinterpret_cast<Quad<E>*>(lds + off) = sh;
        *reinterpret_cast<Quad<E>*>(lds + pipe::PSTR + off) = sl;
    }
    __device__ __forceinline__ void whole(const f32x16& a, int p, int g)
    {
        load(p, g);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int i = 0; i < 4; ++i) stage(a, g, st, i);
        store();
    }
};

// K loop over the image whose first absolute row is row_base.  SHADOW: retire epilogue 2 of the previous board
// (accumulators prev[3]) while the MFMAs run.
template <typename E, bool SHADOW>
__device__ __forceinline__ void pipe_kloop(unsigned char* lds, int row_base, const uint4* wq, int lane, f32x16* acc,
                                           f32x16* prev, PipeShadow<E>& shd)
{
    using namespace pipe;
    typedef typename Mfma<E>::V8 V8;
    const int kb = lane >> 5, ln = lane & 31;
    int pre[NT], pre_n[NT];
    int qy[3], qx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = t * 32 + ln;
        qy[t] = q < 90 ? q / 9 : 100;
        qx[t] = q - (q / 9) * 9;
    }
    auto tap_row = [&](int dy, int dx, int t) {
        const bool ok = (unsigned)(qy[t] + dy) < 10u && (unsigned)(qx[t] + dx) < 9u;
        const int nominal = t * 32 + ln + dy * 9 + dx;
        const int row = ok ? row_base + nominal : ROW_Z + (nominal & 15);
        return row * RB + (((kb ^ nominal) & 15) << 4);       // swizzle key = the image-relative row
    };
    V8 wf[W_RING][2];
    V8 px[2][NT][2];
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    auto load_w = [&](int step, int part) {
        return __builtin_bit_cast(V8, wq[(size_t)part * W_PART + (size_t)step * W_STEP]);
    };
    auto load_px = [&](int off, int part) {
        return __builtin_bit_cast(V8, *reinterpret_cast<const uint4*>(lds + part * PSTR + off));
    };
#pragma unroll
    for (int p = 0; p < NT; ++p) pre[p] = tap_row(-1, -1, p);
#pragma unroll
    for (int s = 0; s < W_RING - 1; ++s)
#pragma unroll
        for (int part = 0; part < 2; ++part) wf[s][part] = load_w(s, part);
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int p = 0; p < NT; ++p) px[0][p][part] = load_px(pre[p], part);

    constexpr int NL = NT * 2, PER = 1;
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {                 // taps 3j .. 3j+2 (dy = j - 1); shadow: the 4 units of pixel tile j
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int tap = 3 * j + tt;
            const int ndy = tt < 2 ? j - 1 : (j < 2 ? j : 1), ndx = tt < 2 ? tt : -1;   // the NEXT tap (last: itself)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int step = tap * KK + kk;
                const V8* w = wf[kk % W_RING];
                V8 (*b)[2] = px[kk & 1];
                V8 (*bn)[2] = px[(kk + 1) & 1];
                const int* rows = kk + 1 < KK ? pre : pre_n;
                const int kn = (kk + 1) % KK;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const int pass = i / NT, p = i % NT;
                    acc[p] = Mfma<E>::mma(w[pass == 1 ? 1 : 0], b[p][pass == 2 ? 1 : 0], acc[p]);
                    if (i < NL) bn[i % NT][i / NT] = load_px(rows[i % NT] ^ (kn << 5), i / NT);
                    if (i >= NM - PER && kk * PER + (i - (NM - PER)) < NT)
                        pre_n[kk * PER + (i - (NM - PER))] = tap_row(ndy, ndx, kk * PER + (i - (NM - PER)));
                    if (i >= NM - 2)
                        wf[(kk + W_RING - 1) % W_RING][i - (NM - 2)] = load_w(step + W_RING - 1, i - (NM - 2));
                    if (SHADOW) {
                        // slot fs of 216 in this pass; unit g = fs / 54 of pixel tile j: LDS reads early, the
                        // arithmetic spread over a few slots, the in-place stores late
                        const int fs = (tt * KK + kk) * NM + i, g = fs / 54, r = fs % 54;
                        if (r == 2) shd.load(j, g);
                        if (r >= 12 && r < 44 && (r & 1) == 0) shd.stage(prev[0], g, (r - 12) / 8, ((r - 12) / 2) % 4);
                        if (r == 48) shd.store();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) pre[p] = pre_n[p];
        }
        if (SHADOW) {                              // rotate the retired tile out: the body always reads prev[0]
            prev[0] = prev[1];
            prev[1] = prev[2];
        }
    }
}

}  // namespace
