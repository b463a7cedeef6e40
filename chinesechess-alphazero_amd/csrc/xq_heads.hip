// xq_heads.hip -- the dense tail of the policy / value heads as hand-written MFMA kernels (gfx950).
//
// Reference (cchess_alphazero/agent/model.py:56-66, after the two 1x1 head convolutions):
//     policy:  Flatten -> Dense(2086, softmax)
//     value:   Flatten -> Dense(256, relu) -> Dense(1, tanh)
// Round 2 ran these on hipBLASLt fp32 GEMMs + PyTorch's softmax / clamp / tanh kernels: 0.87 ms of a 31 ms self-play
// round at 32 768 positions (an fp32 GEMM at the fp32 matrix rate, then 546 MB of softmax traffic), eight launches.
// Here:
//   k_fc_tile<POLICY>   logits[n][2086] = feat[n][F] . W^T + b for a tile of 64 positions per workgroup, written
//                       un-normalised, with the row's running (max, sum of exp) kept per lane -- no cross-lane work in
//                       the loop: the positions are the COLUMNS of the MFMA, so a lane owns one position
//   k_policy_normalize  p = exp(logit - max) / sum in place (one streaming pass, rows of 8344 bytes, 8-byte accesses)
//   k_fc_tile<VALUE>    hidden = relu(feat . W1^T + b1) stays in the accumulators, value = tanh(hidden . w2 + b2)
// Arithmetic: the same split-operand scheme as the tower (xq_conv.hip): every fp32 operand is a (hi, lo) pair of 2-byte
// values, a product is accumulated as w_hi x_hi + w_lo x_hi + w_hi x_lo in fp32 by v_mfma_f32_32x32x16_{bf16,f16}.
// dtype CZ_BF16: dropped term and representation error 2^-17 relative per product; CZ_F16 (round 4, what the network
// uses unless bf16x3 was asked for): (hi, lo) fp16 pairs hold 22 bits where fp16's range holds the lo part (features are
// O(1), fp16 subnormal inputs are honoured by the matrix unit -- tools/f16x3_probe.py), 2^-21 class, same cost.  Order: K-steps of 16 features in index
// order, the three terms of a step in the order above; the softmax statistics are accumulated per lane over its label
// tiles in index order, combined across the two half-waves and the four waves of a position tile in a fixed order.
// Weights are packed on the host in fragment order (cz_fc_pack_weights); they stream from L2 (3 MB for the policy
// layer), the features of the tile sit in LDS as the B operand.
// The compact evaluation queue's device-side count is honoured (n_dev): rows beyond it are not computed or written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/czero.h"

extern "C" void czi_set_error(const char* msg);

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename E> struct FcMma;
template <> struct FcMma<__bf16> {
    static __device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct FcMma<_Float16> {
    static __device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int TILE_ROWS = 64;         // positions per workgroup pass (two MFMA column tiles)
constexpr int FC_THREADS = 512;       // 8 waves: wave w -> position tile w & 1, label tiles (w >> 1) mod 4
constexpr int FC_PAD_STEPS = 2;       // zero K-steps appended per label tile (the prefetch reads one step ahead)
constexpr int FC_MAX_KSTEPS = 24;     // features per position <= 384 (the reference's heads: 2 or 4 filters x 90 squares)
constexpr int FC_IMG_BYTES = 2 * TILE_ROWS * (FC_MAX_KSTEPS * 32 + 16);
constexpr int FC_RED_BYTES = 2 * 256 * (int)sizeof(float);
constexpr int FC_STAGE_ROW = 144;     // bytes per position row of a wave's logit staging tile (32 floats + 16 B: odd in 16-B units)
constexpr int FC_STAGE_BYTES = 32 * FC_STAGE_ROW;            // per wave
constexpr int FC_LDS_BYTES = FC_IMG_BYTES + FC_RED_BYTES + (FC_THREADS / 64) * FC_STAGE_BYTES;

enum FcMode { FC_POLICY = 0, FC_VALUE = 1 };

struct FcArgs {
    const float* feat;        // [n][F]
    const void* wp;           // packed weights [label tiles][ksteps + pad][2 parts][64 lanes][8]
    const float* bias;        // [n_out]
    int F, ksteps, n_out, n_tiles;
    // policy
    float* logits;            // [n][n_out]
    float2* stats;            // [n] (max, sum exp)
    // value
    const float* w2;          // [n_out]
    float b2;
    float* value;             // [n]
};

// F: features per position (180 or 360: 2 or 4 head filters x 90 squares; compile-time so that the staging loop divides
// by a constant and the K loop unrolls around a weight ring)
// exp for the running softmax statistics: v_exp_f32 (2^x) on x * log2(e) -- 2 instructions instead of libm's ~25; the
// statistics only scale the row (relative error ~1e-6 of the sum), the probabilities themselves are formed with expf in
// k_policy_normalize.  Half of this kernel's issue slots went to libm exps before (0.277 -> see profiles/r03_*).
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

template <int MODE, int F, typename E>
__global__ __launch_bounds__(FC_THREADS) void k_fc_tile(FcArgs a, int n, const int32_t* __restrict__ n_dev)
{
    constexpr int KS = (F + 15) / 16;                 // K-steps
    constexpr int RB = KS * 32 + 16;                  // bytes per position row of the feature image; RB / 16 is odd, so
    constexpr int PART = TILE_ROWS * RB;              // the 16 lanes of a ds_read_b128 group (16 consecutive positions, same
    constexpr int RING = 6;                           // chunk) fall on 16 different 16-byte slots of the 256-byte bank row
    static_assert(F % 4 == 0 && KS <= FC_MAX_KSTEPS, "whole float4s per row");
    __shared__ __attribute__((aligned(16))) unsigned char lds[FC_LDS_BYTES];
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n = nd < n ? nd : n;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 31, kb = lane >> 5;
    const int bt = wave & 1, lq = wave >> 1;
    float* red = reinterpret_cast<float*>(lds + FC_IMG_BYTES);   // [2 values][2 position tiles][4 wave classes][32]
    unsigned char* stg = lds + FC_IMG_BYTES + FC_RED_BYTES + wave * FC_STAGE_BYTES;      // this wave's logit tile
    const int n_row_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    struct alignas(8) Q4 { E e[4]; };
    for (int rt = blockIdx.x; rt < n_row_tiles; rt += gridDim.x) {
        const int row0 = rt * TILE_ROWS;
        // ---- stage the tile's features as (hi, lo) bf16: the tile is one contiguous block of 64 x F floats ----
        {
            const float4* src = reinterpret_cast<const float4*>(a.feat + (size_t)row0 * F);
            const int rows_here = n - row0 < TILE_ROWS ? n - row0 : TILE_ROWS;
            for (int i = tid; i < TILE_ROWS * (F / 4); i += FC_THREADS) {
                const int r = i / (F / 4), k = (i - r * (F / 4)) * 4;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (r < rows_here) v = src[i];
                const float f[4] = {v.x, v.y, v.z, v.w};
                Q4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hi.e[j] = (E)f[j];
                    lo.e[j] = (E)(f[j] - (float)hi.e[j]);
                }
                *reinterpret_cast<Q4*>(lds + r * RB + k * 2) = hi;
                *reinterpret_cast<Q4*>(lds + PART + r * RB + k * 2) = lo;
            }
            if (KS * 16 > F) {                        // zero the K padding of the last step
                constexpr int PADQ = (KS * 16 - F) / 4;
                for (int i = tid; i < TILE_ROWS * PADQ; i += FC_THREADS) {
                    const int r = i / PADQ, k = F + (i - r * PADQ) * 4;
                    const Q4 z{};
                    *reinterpret_cast<Q4*>(lds + r * RB + k * 2) = z;
                    *reinterpret_cast<Q4*>(lds + PART + r * RB + k * 2) = z;
                }
            }
        }
        __syncthreads();

        const unsigned char* brow = lds + (bt * 32 + ln) * RB + kb * 16;
        float m_run = -3.0e38f, s_run = 0.0f;      // policy: running max / sum of exp of this lane's labels
        float dot = 0.0f;                          // value: partial hidden . w2
        const int board = row0 + bt * 32 + ln;
        // two label tiles per pass and wave (two independent MFMA chains, one set of position operands), weights through a
        // register ring RING K-steps deep: an L2 round trip is several hundred cycles, a K-step of one tile only 96
        for (int lt0 = lq * 2; lt0 < a.n_tiles; lt0 += 8) {
            const bool two = lt0 + 1 < a.n_tiles;
            const uint4* wq0 = reinterpret_cast<const uint4*>(a.wp) + (size_t)lt0 * (KS + FC_PAD_STEPS) * 128 + lane;
            const uint4* wq1 = wq0 + (two ? (size_t)(KS + FC_PAD_STEPS) * 128 : 0);
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
            uint4 w[RING][4];                      // [ring slot][tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo]
#pragma unroll
            for (int s = 0; s < RING - 1; ++s) {
                if (s < KS) {
                    w[s][0] = wq0[(size_t)s * 128]; w[s][1] = wq0[(size_t)s * 128 + 64];
                    w[s][2] = wq1[(size_t)s * 128]; w[s][3] = wq1[(size_t)s * 128 + 64];
                }
            }
            uint4 xh = *reinterpret_cast<const uint4*>(brow), xl = *reinterpret_cast<const uint4*>(brow + PART);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + RING - 1 < KS) {
                    const int sl = (ks + RING - 1) % RING;
                    const size_t o = (size_t)(ks + RING - 1) * 128;
                    w[sl][0] = wq0[o]; w[sl][1] = wq0[o + 64]; w[sl][2] = wq1[o]; w[sl][3] = wq1[o + 64];
                }
                uint4 xh_n = xh, xl_n = xl;
                if (ks + 1 < KS) {
                    xh_n = *reinterpret_cast<const uint4*>(brow + (ks + 1) * 32);
                    xl_n = *reinterpret_cast<const uint4*>(brow + PART + (ks + 1) * 32);
                }
                const uint4* c = w[ks % RING];
                acc0 = FcMma<E>::mma(c[0], xh, acc0);
                acc1 = FcMma<E>::mma(c[2], xh, acc1);
                acc0 = FcMma<E>::mma(c[1], xh, acc0);
                acc1 = FcMma<E>::mma(c[3], xh, acc1);
                acc0 = FcMma<E>::mma(c[0], xl, acc0);
                acc1 = FcMma<E>::mma(c[2], xl, acc1);
                xh = xh_n; xl = xl_n;
                __builtin_amdgcn_sched_barrier(0);       // keep the loads where they are issued: RING - 1 steps ahead
            }
            // acc[4 g + i] <-> label lt * 32 + 8 g + 4 kb + i of position `board`
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
                if (tile == 1 && !two) break;
                const int lt = lt0 + tile;
                const f32x16& acc = tile ? acc1 : acc0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lab = lt * 32 + g * 8 + kb * 4;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = lab + i < a.n_out ? acc[g * 4 + i] + a.bias[lab + i] : 0.0f;
                    if (MODE == FC_POLICY) {
                        float mx = m_run;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (lab + i < a.n_out) mx = v[i] > mx ? v[i] : mx;
                        float s = s_run * fexp(m_run - mx);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (lab + i < a.n_out) s += fexp(v[i] - mx);
                        m_run = mx;
                        s_run = s;
                        // into the wave's staging tile [position][32 labels]; written out in whole rows below
                        *reinterpret_cast<float4*>(stg + ln * FC_STAGE_ROW + (g * 8 + kb * 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (lab + i < a.n_out) {
                                const float h = v[i] > 0.0f ? v[i] : 0.0f;
                                dot += h * a.w2[lab + i];
                            }
                    }
                }
                if (MODE == FC_POLICY) {
                    // the tile leaves through LDS so that HBM sees 128 contiguous bytes per position (8 lanes x 16 B) instead of
                    // 8-byte pieces from the accumulator layout (34 M write requests per launch: a request-rate limit)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = (lane >> 3) + 8 * j, c4 = (lane & 7) * 4;
                        const float4 o = *reinterpret_cast<const float4*>(stg + r * FC_STAGE_ROW + c4 * 4);
                        const int brd = row0 + bt * 32 + r, l0 = lt * 32 + c4;
                        if (brd < n) {
                            float* dst = a.logits + (size_t)brd * a.n_out + l0;      // 8-byte aligned (n_out is even)
                            if (l0 + 3 < a.n_out) {
                                *reinterpret_cast<float2*>(dst) = make_float2(o.x, o.y);
                                *reinterpret_cast<float2*>(dst + 2) = make_float2(o.z, o.w);
                            } else {
                                if (l0 < a.n_out) dst[0] = o.x;
                                if (l0 + 1 < a.n_out) dst[1] = o.y;
                                if (l0 + 2 < a.n_out) dst[2] = o.z;
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        // ---- combine: the two half-waves (kb), then the four waves that share the position tile ----
        if (MODE == FC_POLICY) {
            const float m_o = __shfl_xor(m_run, 32, 64), s_o = __shfl_xor(s_run, 32, 64);
            const float mx = m_run > m_o ? m_run : m_o;
            const float s = s_run * fexp(m_run - mx) + s_o * fexp(m_o - mx);
            if (kb == 0) {
                red[(bt * 4 + lq) * 32 + ln] = mx;
                red[256 + (bt * 4 + lq) * 32 + ln] = s;
            }
        } else {
            const float d = dot + __shfl_xor(dot, 32, 64);
            if (kb == 0) red[(bt * 4 + lq) * 32 + ln] = d;
        }
        __syncthreads();
        if (lq == 0 && kb == 0 && board < n) {
            if (MODE == FC_POLICY) {
                float mx = red[(bt * 4) * 32 + ln];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float t = red[(bt * 4 + q) * 32 + ln];
                    mx = t > mx ? t : mx;
                }
                float s = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    s += red[256 + (bt * 4 + q) * 32 + ln] * fexp(red[(bt * 4 + q) * 32 + ln] - mx);
                a.stats[board] = make_float2(mx, s);
            } else {
                float d = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) d += red[(bt * 4 + q) * 32 + ln];
                a.value[board] = tanhf(d + a.b2);
            }
        }
        __syncthreads();                    // the image and `red` are reused by the next tile
    }
}

// p = exp(logit - max) / sum, in place: one wave per row (n_out floats, n_out even: 8-byte accesses stay aligned)
__global__ __launch_bounds__(256) void k_policy_normalize(float* __restrict__ p, const float2* __restrict__ stats, int n,
                                                         int n_out, const int32_t* __restrict__ n_dev)
{
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n = nd < n ? nd : n;
    }
    const int half = n_out / 2, lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) {
        const float2 st = stats[row];
        float2* r = reinterpret_cast<float2*>(p + (size_t)row * n_out);
        for (int i = lane; i < half; i += 64) {
            float2 v = r[i];
            v.x = expf(v.x - st.x) / st.y;
            v.y = expf(v.y - st.x) / st.y;
            r[i] = v;
        }
    }
}

inline uint16_t host_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float host_bf16_f(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline uint16_t host_f16(float f)
{
    const _Float16 h = (_Float16)f;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
inline float host_f16_f(uint16_t b)
{
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

int device_cus()
{
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        n_cu = prop.multiProcessorCount;
    }
    return n_cu;
}

}  // namespace

extern "C" size_t cz_fc_packed_elems(int n_out, int n_in)
{
    if (n_out < 1 || n_in < 1 || n_in > 2048 || n_out > 65536) return 0;
    const size_t tiles = (size_t)(n_out + 31) / 32, ksteps = (size_t)(n_in + 15) / 16;
    return tiles * (ksteps + FC_PAD_STEPS) * 2 * 64 * 8;
}

extern "C" int cz_fc_pack_weights(const float* w, int n_out, int n_in, int dtype, void* out_host)
{
    const size_t elems = cz_fc_packed_elems(n_out, n_in);
    if (!w || !out_host || elems == 0 || (dtype != CZ_BF16 && dtype != CZ_F16)) {
        czi_set_error("cz_fc_pack_weights: bad argument (dtype: CZ_BF16 or CZ_F16 pairs)");
        return CZ_ERR_ARG;
    }
    uint16_t* out = (uint16_t*)out_host;
    memset(out, 0, elems * sizeof(uint16_t));
    const int tiles = (n_out + 31) / 32, ksteps = (n_in + 15) / 16;
    for (int lt = 0; lt < tiles; ++lt)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = lt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
                    if (o >= n_out || k >= n_in) continue;
                    const float v = w[(size_t)o * n_in + k];
                    const uint16_t hi = dtype == CZ_BF16 ? host_bf16(v) : host_f16(v);
                    const size_t base = (((size_t)lt * (ksteps + FC_PAD_STEPS) + ks) * 2) * 64 * 8;
                    out[base + (size_t)lane * 8 + j] = hi;
                    out[base + 64 * 8 + (size_t)lane * 8 + j] =
                        dtype == CZ_BF16 ? host_bf16(v - host_bf16_f(hi)) : host_f16(v - host_f16_f(hi));
                }
    return CZ_OK;
}

extern "C" int cz_heads_tail(const float* policy_feat, int n_policy_feat, const void* wp_packed, const float* bias_p,
                             int n_labels, const float* value_feat, int n_value_feat, const void* w1_packed,
                             const float* bias1, int n_hidden, const float* w2, float b2, float* policy, float* value,
                             float* stats_scratch, int n_boards, int dtype, int normalize, const int32_t* n_dev, void* stream)
{
    if (!policy_feat || !wp_packed || !bias_p || !value_feat || !w1_packed || !bias1 || !w2 || !policy || !value ||
        !stats_scratch || n_boards < 0 || (dtype != CZ_BF16 && dtype != CZ_F16) || n_labels < 2 || (n_labels & 1) || n_hidden < 1 || n_policy_feat < 1 ||
        (n_policy_feat != 180 && n_policy_feat != 360) || (n_value_feat != 180 && n_value_feat != 360)) {
        czi_set_error("cz_heads_tail: bad argument (n_labels even; 180 or 360 features per head: 2 or 4 filters x 90 squares; "
                      "dtype of the packed pairs: CZ_BF16 or CZ_F16)");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = device_cus();
    if (n_cu < 0) {
        czi_set_error("cz_heads_tail: cannot query the device");
        return CZ_ERR_HIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int row_tiles = (n_boards + TILE_ROWS - 1) / TILE_ROWS;
    {
        FcArgs a{};
        a.feat = policy_feat; a.wp = wp_packed; a.bias = bias_p; a.F = n_policy_feat;
        a.ksteps = (n_policy_feat + 15) / 16; a.n_out = n_labels; a.n_tiles = (n_labels + 31) / 32;
        a.logits = policy; a.stats = reinterpret_cast<float2*>(stats_scratch);
        const unsigned blocks = (unsigned)(row_tiles < 2 * n_cu ? row_tiles : 2 * n_cu);
#define CZ_FC(MODE, FEAT)                                                                                              \
        do {                                                                                                          \
            if (dtype == CZ_BF16) hipLaunchKernelGGL((k_fc_tile<MODE, FEAT, __bf16>), dim3(blocks), dim3(FC_THREADS), 0, st, a, n_boards, n_dev); \
            else hipLaunchKernelGGL((k_fc_tile<MODE, FEAT, _Float16>), dim3(blocks), dim3(FC_THREADS), 0, st, a, n_boards, n_dev);                \
        } while (0)
        if (n_policy_feat == 360) CZ_FC(FC_POLICY, 360);
        else CZ_FC(FC_POLICY, 180);
        if (normalize) {              // (0: `policy` keeps the raw logits -- cz_search_policy_logits takes them as they are)
            size_t nb = ((size_t)n_boards + 3) / 4;
            if (nb > (size_t)n_cu * 16) nb = (size_t)n_cu * 16;
            hipLaunchKernelGGL(k_policy_normalize, dim3((unsigned)nb), dim3(256), 0, st, policy,
                               reinterpret_cast<const float2*>(stats_scratch), n_boards, n_labels, n_dev);
        }
    }
    {
        FcArgs a{};
        a.feat = value_feat; a.wp = w1_packed; a.bias = bias1; a.F = n_value_feat;
        a.ksteps = (n_value_feat + 15) / 16; a.n_out = n_hidden; a.n_tiles = (n_hidden + 31) / 32;
        a.w2 = w2; a.b2 = b2; a.value = value;
        const unsigned blocks = (unsigned)(row_tiles < 2 * n_cu ? row_tiles : 2 * n_cu);
        if (n_value_feat == 180) CZ_FC(FC_VALUE, 180);
        else CZ_FC(FC_VALUE, 360);
#undef CZ_FC
    }
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_heads_tail: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}
