// xq_tower.hip -- CHAINS of residual blocks of the 128-filter tower in one launch (gfx950), for every tower arithmetic.
//
// Reference: the residual tower of CChessModel.build (cchess_alphazero/agent/model.py:41-43, `for _ in range(res_layer_num):
// x = self._build_residual_block(x)`, blocks at :68-83) evaluated by the prediction thread (agent/api.py:63-64).
//
// Round 5 chained the c6 blocks (k_tower_c6): a workgroup takes a PAIR of boards through all blocks of the chain with the
// activations staying in LDS (three images XA | XB | Y sharing 16 zero rows), so HBM sees a board at the chain's entry and exit
// only.  Round 6 gives the other arithmetics the same chain -- a peaked-policy network (what training
// produces) is sent to c8 / f16x3 blocks by the load-time guard and ran one launch per block:
//   k_tower<HEADS, UNI>       blocks whose result is STAGED in fp32 and converted by the copy waves, all of one image format
//                             (UNI = CZ_IMG_C6: round 5's kernel; CZ_IMG_C8).  Exit: the operand triple to HBM -- a c6 chain may
//                             end on the block that writes a c8 image (the hand-over of a c6>N tower) --, (hi, lo) fp16 pairs
//                             (the hand-over of a c8>N tower to its f16x3 blocks), or the 1 x 1 head convolutions.  (A form with
//                             per-block formats chosen at run time -- a whole hybrid tower in one launch -- was built and
//                             dropped: its c8 blocks ran 6 % SLOWER than one launch per block, 107 spilled registers.)
//   k_tower_pairs<E, HEADS>   (hi, lo) pair blocks (f16x3 / bf16x3): k_resblock_pipe's schedule -- the second epilogue of
//                             the previous step runs IN PLACE in the shadow of K loop 1 -- which already leaves a block's
//                             result in the operand layout in the dead image: the chain needs no conversion at all.
// Arithmetic, accumulation order and conversions are those of the one-block kernels (k_resblock_c8<.., C6>, k_resblock_c8,
// k_resblock_pipe): a chain is BIT-IDENTICAL to block-by-block launches (tests/test_gpu_c6.py, tests/test_gpu_tower.py); the
// HEADS exits sum a pixel's head dot products over four 32-channel partial sums (float32 rounding of that order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/czero.h"
#include "xq_c8_kloop.h"
#include "xq_nn_common.h"

extern "C" void czi_set_error(const char* msg);
// csrc/xq_conv.hip: cz_tower's launches on the four-wave pair kernel (k_resblock_ip4_c8<128>)
extern "C" int czi_tower4_launch(const void* x_hi, const void* x_img, int n_blocks, const void* const* w1, const float* const* b1,
                                 const void* const* w2, const float* const* b2, int c6, int exit_mode, void* y_hi, void* y_img,
                                 const float* head_w, const float* head_b, float* pol, float* val, int n_pol, int n_boards,
                                 int n_cu, const int32_t* n_dev, void* stream);

namespace {

typedef c8k::u32x4 u32x4;

namespace tw {
constexpr int C = 128, RB = 256, ROW_XA = 0, ROW_XB = 90, ROW_Y = 180, ROW_DUMP = 270, ROW_Z = 272, PSTR = (ROW_Z + 16) * RB;
constexpr int BIAS_OFF = 2 * PSTR;                 // float bias[2 buffers][2 convolutions][128]
constexpr int HW_OFF = BIAS_OFF + 2 * 2 * C * 4;   // HEADS: float head_w[6][128]
constexpr int LDS_BYTES = HW_OFF, LDS_BYTES_HEADS = HW_OFF + 6 * C * 4;
constexpr int MAX_BLOCKS = 8;
static_assert(LDS_BYTES_HEADS <= 160 * 1024, "XA + XB + Y + zero rows + bias buffers (+ head filters) must fit the CU's LDS");
static_assert(ROW_Z == pipe::ROW_Z && PSTR == pipe::PSTR && RB == pipe::RB, "k_tower_pairs runs pipe_kloop on this layout");
// image formats (include/czero.h CZ_IMG_*)
constexpr int F_C8 = CZ_IMG_C8, F_C6 = CZ_IMG_C6, F_PAIR = CZ_IMG_PAIR;
struct Chain {
    const void* w1[MAX_BLOCKS];
    const void* w2[MAX_BLOCKS];
    const float* b1[MAX_BLOCKS];
    const float* b2[MAX_BLOCKS];
    int n;
    int exit_fmt;                       // format of the image the chain's exit stores (ignored by HEADS)
    unsigned char fx[MAX_BLOCKS];       // format of the image a block's first convolution reads (= what the block before wrote)
    unsigned char fy[MAX_BLOCKS];       // ... of the block's intermediate image (its second convolution's operands)
};
// fp32 staging inside an image: byte offset (from the image's first row in the f16 part) of channels ch .. ch + 3 of pixel q
__device__ __forceinline__ int stage_off(int q, int ch) { return (ch >> 6) * PSTR + q * RB + ((((ch >> 2) & 15) ^ (q & 15)) << 4); }

struct Shadow {                         // relu(acc2 of the previous step) -> staging inside the previous step's own X image
    unsigned char* lds;
    f32x16* prev;
    int wave, kb, ln, base;             // base: byte offset of that image's first row
    __device__ __forceinline__ void unit(const f32x16& a, int q, int g)
    {
        const int ch = wave * 32 + g * 8 + kb * 4;
        const int addr = q < 90 ? base + stage_off(q, ch) : ROW_DUMP * RB + (kb * 6 + (ln - 26)) * 16;
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
}  // namespace tw

// ---- kernel: a chain of STAGED blocks (c6 / c8 images) ------------------------------------------------------------------------
// The steps of a pair of boards are (block 0, A), (block 0, B), (block 1, A), ...; a step is a board of k_resblock_c8: K loop 1
// on X[s] with the PREVIOUS step's second epilogue in its shadow (relu -> fp32 staging inside the other slot's dead image),
// epilogue 1 -> Y (accumulators restart at b2 + skip), K loop 2.  During K loop 2 the copy waves turn the staged result into
// the operand triple of the NEXT block's format in place (four adjacent lanes per pixel row, all LDS reads of a wave before its
// writes, every pass on one (row, 32-channel block) mapping: no synchronisation among the copy waves).  Per-block filters,
// biases (double-buffered in LDS), formats and image exponents switch every two steps; a workgroup with an odd number of
// boards runs its last board in both slots (the copy is not stored).
template <bool HEADS, int UNI>
__global__ __launch_bounds__(512, 2) void k_tower(
    const _Float16* __restrict__ xh, const unsigned char* __restrict__ xc, tw::Chain ch, _Float16* __restrict__ yh,
    unsigned char* __restrict__ yc, int n_boards, const int32_t* __restrict__ n_dev, HeadArgs hd)
{
    using namespace tw;
    using rb8::c6_chunk;
    using rb8::c6_lds_off;
    using rb8::pack_ints;
    using rb8::u32x6;
    using rb8::f32x32;
    static_assert(CZ_C6_TAIL_SWZ == 0, "the chained tower keeps one tail convention");
    constexpr int NT = 3, CTHR = 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HEADS ? LDS_BYTES_HEADS : LDS_BYTES];
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stride = gridDim.x, t0 = blockIdx.x;
    if (t0 >= n_boards) return;
    const int NB = ch.n;
    const int mine = (n_boards - t0 + stride - 1) / stride;          // boards of this workgroup
    const int pairs = (mine + 1) / 2;
    // board of (pair p, slot s); the last board twice when the count is odd (`real` false for the copy: not stored)
    auto board_of = [&](int p, int s, bool& real) __attribute__((always_inline)) {
        const int k = 2 * p + s;
        real = k < mine;
        return t0 + (real ? k : mine - 1) * stride;
    };
    auto row_of = [](int s) { return s ? ROW_XB : ROW_XA; };
    static_assert(UNI == F_C8 || UNI == F_C6, "one image format per chain");
    auto fx_of = [&](int) { return UNI; };
    auto fy_of = [&](int) { return UNI; };
    // the format block b's result is converted to: the next block's input, or the exit's (a c6 chain may end on the block that
    // hands a c8 image over to c8 blocks, a c8 chain on the one that hands fp16 pairs over to f16x3 blocks)
    auto fo_of = [&](int b) { return b + 1 < NB ? UNI : ch.exit_fmt; };

    if (wave >= 4) {                                    // ---- copy waves ----
        const int ctid = tid - 256;
        // Every pass of the copy waves uses ONE mapping: item i = it * 256 + ctid < 360 is (pixel row i >> 2, 32-channel block
        // i & 3) -- the row's f16 chunks 4 blk .. 4 blk + 3 and four chunks of the correction image's row (c6: the block's two
        // pieces, head + tail chunk each; c8: the block's 32 lo bytes and 32 value bytes).  A thread only ever touches the bytes
        // of its own row quarter, and the four quarters of a row sit in adjacent lanes of one wave, so load -> write, convert
        // (in place) and drain need no synchronisation among the copy waves.
        // entry: HBM -> X[s], the image rows copied chunk for chunk (any format: 16 chunks of the f16 row, 16 of the image row).
        // The loads are issued and consumed inside ONE window (under K loop 2: ~9 us for a ~2 us round trip), so their 64
        // registers are never live across convert() -- round 5 loaded a window earlier and the compiler spilled them:
        // 47 VGPRs through scratch per board, which is where that build's 2x WRITE_SIZE came from (EXPERIMENTS round 6).
        auto fill_x = [&](int board, int s) __attribute__((always_inline)) {
            const u32x4* sh = reinterpret_cast<const u32x4*>(xh + (size_t)board * 90 * C);
            const u32x4* sc = reinterpret_cast<const u32x4*>(xc + (size_t)board * 90 * 2 * C);
            unsigned char* X = lds + row_of(s) * RB;
            u32x4 v[2][8];                              // (a native vector type: HIP's uint4 is a struct whose copies become
                                                         //  memcpy calls the compiler leaves in SCRATCH -- global -> scratch -> LDS)
            // (the sources are read-only __restrict__ pointers: without a dependence the compiler hoists these loads above the
            //  conversion that precedes them -- and spills what they fetched)
            int ct2 = ctid;
            asm volatile("" : "+v"(ct2) :: "memory");
            // (the lanes without an item of their own -- ctid >= 104 in the second pass -- load item 359 again, so that every
            //  element of v[][] is defined on every path, and write nothing)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i0 = it * CTHR + ct2, i = i0 < 90 * 4 ? i0 : 90 * 4 - 1;
                const int qq = i >> 2, blk = i & 3, c0 = c6_chunk(0, blk);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[it][k] = sh[qq * 16 + blk * 4 + k];
                v[it][4] = sc[qq * 16 + c0];
                v[it][5] = sc[qq * 16 + c0 + 1];
                v[it][6] = sc[qq * 16 + 8 + c0];
                v[it][7] = sc[qq * 16 + 8 + c0 + 1];
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i0 = it * CTHR + ct2, i = i0 < 90 * 4 ? i0 : 90 * 4 - 1;
                if (i0 >= 90 * 4) continue;               // (a lane without an item must not write: row 89 belongs to another WAVE, which
                                                          //  may still be reading its staging -- the rows are the only synchronisation)
                const int qq = i >> 2, blk = i & 3, c0 = c6_chunk(0, blk);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<u32x4*>(X + qq * RB + (((blk * 4 + k) ^ (qq & 15)) << 4)) = v[it][k];
                *reinterpret_cast<u32x4*>(X + PSTR + c6_lds_off(qq, c0)) = v[it][4];
                *reinterpret_cast<u32x4*>(X + PSTR + c6_lds_off(qq, c0 + 1)) = v[it][5];
                *reinterpret_cast<u32x4*>(X + PSTR + c6_lds_off(qq, 8 + c0)) = v[it][6];
                *reinterpret_cast<u32x4*>(X + PSTR + c6_lds_off(qq, 8 + c0 + 1)) = v[it][7];
            }
        };
        struct alignas(16) H8 { Quad<_Float16> a, b; };
        // fp32 staging of slot s -> its c6 operand triple, in place (k_out: the exponent of the image, from the block that made
        // it); board >= 0: the chain's exit -- the triple also goes to HBM, straight from the registers
        auto convert_c6 = [&](int s, int k_out, int board) __attribute__((always_inline)) {
            unsigned char* X = lds + row_of(s) * RB;
            const float s_hi = __builtin_ldexpf(1.0f, k_out), s_lo = __builtin_ldexpf(1.0f, k_out - cf8::X_LO_SHIFT);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, blk = i & 3;
                f32x16 av, bv, al, bl;
                H8 h[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 f0 = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 8 * k));
                    const float4 f1 = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 8 * k + 4));
                    const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        h[k].a.e[j] = (_Float16)r[j];
                        h[k].b.e[j] = (_Float16)r[4 + j];
                        av[4 * k + j] = r[j];
                        bv[4 * k + j] = r[4 + j];
                        al[4 * k + j] = r[j] - (float)h[k].a.e[j];
                        bl[4 * k + j] = r[4 + j] - (float)h[k].b.e[j];
                    }
                }
                const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
                const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
                // (every lane of the wave has read its part of the row by now: the writes below may land on bytes another
                //  lane of the SAME row -- one of the three neighbours in this wave -- has just read)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<uint4*>(X + qq * RB + (((blk * 4 + k) ^ (qq & 15)) << 4)) = __builtin_bit_cast(uint4, h[k]);
                unsigned char* P1 = X + PSTR;
                *reinterpret_cast<uint4*>(P1 + c6_lds_off(qq, c6_chunk(0, blk))) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                *reinterpret_cast<uint2*>(P1 + c6_lds_off(qq, c6_chunk(0, blk) + 1)) = make_uint2(pl[4], pl[5]);
                *reinterpret_cast<uint4*>(P1 + c6_lds_off(qq, c6_chunk(1, blk))) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                *reinterpret_cast<uint2*>(P1 + c6_lds_off(qq, c6_chunk(1, blk) + 1)) = make_uint2(pv[4], pv[5]);
                if (board >= 0) {
                    const size_t ebase = (size_t)board * 90 * C;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        reinterpret_cast<uint4*>(yh + ebase)[qq * 16 + blk * 4 + k] = __builtin_bit_cast(uint4, h[k]);
                    unsigned char* row = yc + ebase * 2 + (size_t)qq * 2 * C;
                    *reinterpret_cast<uint4*>(row + 16 * c6_chunk(0, blk)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                    *reinterpret_cast<uint2*>(row + 16 * c6_chunk(0, blk) + 16) = make_uint2(pl[4], pl[5]);
                    *reinterpret_cast<uint4*>(row + 16 * c6_chunk(1, blk)) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                    *reinterpret_cast<uint2*>(row + 16 * c6_chunk(1, blk) + 16) = make_uint2(pv[4], pv[5]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        };
        // ... -> its c8 operand triple (f16 | e4m3(x_lo 2^11) x 128 | e4m3(x) x 128: k_resblock_c8's store_tile), in place.
        // The item's 32 lo bytes are chunks 2 blk, 2 blk + 1 of the image row, its 32 value bytes chunks 8 + 2 blk, 9 + 2 blk.
        auto convert_c8 = [&](int s, int board) __attribute__((always_inline)) {
            unsigned char* X = lds + row_of(s) * RB;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, blk = i & 3;
                H8 h[4];
                uint32_t l8[8], h8[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 f0 = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 8 * k));
                    const float4 f1 = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 8 * k + 4));
                    const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                    const cf8::Split4 s0 = cf8::split4(r), s1 = cf8::split4(r + 4);
                    h[k].a = s0.hi; h[k].b = s1.hi;
                    l8[2 * k] = s0.l8; l8[2 * k + 1] = s1.l8;
                    h8[2 * k] = s0.h8; h8[2 * k + 1] = s1.h8;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<uint4*>(X + qq * RB + (((blk * 4 + k) ^ (qq & 15)) << 4)) = __builtin_bit_cast(uint4, h[k]);
                unsigned char* P1 = X + PSTR;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    *reinterpret_cast<uint4*>(P1 + c6_lds_off(qq, 2 * blk + j)) = make_uint4(l8[4 * j], l8[4 * j + 1], l8[4 * j + 2], l8[4 * j + 3]);
                    *reinterpret_cast<uint4*>(P1 + c6_lds_off(qq, 8 + 2 * blk + j)) = make_uint4(h8[4 * j], h8[4 * j + 1], h8[4 * j + 2], h8[4 * j + 3]);
                }
                if (board >= 0) {
                    const size_t ebase = (size_t)board * 90 * C;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        reinterpret_cast<uint4*>(yh + ebase)[qq * 16 + blk * 4 + k] = __builtin_bit_cast(uint4, h[k]);
                    uint4* row = reinterpret_cast<uint4*>(yc + ebase * 2 + (size_t)qq * 2 * C);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        row[2 * blk + j] = make_uint4(l8[4 * j], l8[4 * j + 1], l8[4 * j + 2], l8[4 * j + 3]);
                        row[8 + 2 * blk + j] = make_uint4(h8[4 * j], h8[4 * j + 1], h8[4 * j + 2], h8[4 * j + 3]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        };
        // exit of a c8>N tower's c8 part: staging -> (hi, lo) fp16 pairs to HBM (cz_split_bias_act's split: hi = f16(v),
        // lo = f16(v - hi)), y_img = the lo array [n][90][128] f16.  Nothing is written to LDS.
        auto exit_pairs = [&](int s, int board) __attribute__((always_inline)) {
            if (board < 0) return;
            const unsigned char* X = lds + row_of(s) * RB;
            const size_t ebase = (size_t)board * 90 * C;
            _Float16* yl = reinterpret_cast<_Float16*>(yc);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, blk = i & 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 f0 = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 8 * k));
                    const float4 f1 = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 8 * k + 4));
                    const float r[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                    H8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hi.a.e[j] = (_Float16)r[j];
                        hi.b.e[j] = (_Float16)r[4 + j];
                        lo.a.e[j] = (_Float16)(r[j] - (float)hi.a.e[j]);
                        lo.b.e[j] = (_Float16)(r[4 + j] - (float)hi.b.e[j]);
                    }
                    reinterpret_cast<uint4*>(yh + ebase)[qq * 16 + blk * 4 + k] = __builtin_bit_cast(uint4, hi);
                    reinterpret_cast<uint4*>(yl + ebase)[qq * 16 + blk * 4 + k] = __builtin_bit_cast(uint4, lo);
                }
            }
        };
        // HEADS exit: staging of slot s -> the six head features of every pixel of `board` (nothing is written to LDS)
        auto heads_exit = [&](int s, int board) __attribute__((always_inline)) {
            const unsigned char* X = lds + row_of(s) * RB;
            const float* hwl = reinterpret_cast<const float*>(lds + HW_OFF);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, blk = i & 3;
                float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 f = *reinterpret_cast<const float4*>(X + stage_off(qq, blk * 32 + 4 * k));
#pragma unroll
                    for (int o = 0; o < 6; ++o) {
                        const float4 w = *reinterpret_cast<const float4*>(hwl + o * C + blk * 32 + 4 * k);
                        a[o] += f.x * w.x; a[o] += f.y * w.y; a[o] += f.z * w.z; a[o] += f.w * w.w;
                    }
                }
#pragma unroll
                for (int o = 0; o < 6; ++o) {                  // the four lanes of the pixel: (a0 + a1) + (a2 + a3) on every lane
                    a[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[o]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    a[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[o]), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                }
                if (board < 0) continue;
#pragma unroll
                for (int o = 0; o < 6; ++o)
                    if ((o & 3) == blk) {                       // lane blk writes outputs blk and blk + 4
                        float hv = a[o] + hd.b[o];
                        hv = hv > 0.0f ? hv : 0.0f;
                        if (o < hd.n_pol) hd.pol[(size_t)board * (hd.n_pol * 90) + o * 90 + qq] = hv;
                        else hd.val[(size_t)board * ((6 - hd.n_pol) * 90) + (o - hd.n_pol) * 90 + qq] = hv;
                    }
            }
        };
        // the staged result of block pb in slot s -> what comes next (board >= 0: the chain's exit for that board)
        auto retire = [&](int s, int pb, int board) __attribute__((always_inline)) {
            if (HEADS && pb == NB - 1) { heads_exit(s, board); return; }
            const int fo = fo_of(pb);
            if (fo == F_C6) convert_c6(s, __builtin_amdgcn_readfirstlane(pack_ints(ch.w2[pb])[3]), board);
            else if (fo == F_C8) convert_c8(s, board);
            else exit_pairs(s, board);
        };
        auto write_bias = [&](int b) __attribute__((always_inline)) {
            if (ctid < C) {
                float* dst = reinterpret_cast<float*>(lds + BIAS_OFF) + (b & 1) * 2 * C;
                dst[ctid] = ch.b1[b][ctid];
                dst[C + ctid] = ch.b2[b][ctid];
            }
        };
        // ---- prologue: slot A's first board, the zero rows, block 0's bias
        bool real;
        fill_x(board_of(0, 0, real), 0);
        for (int i = ctid; i < 16 * 16; i += CTHR) {
            *reinterpret_cast<uint4*>(lds + ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + PSTR + ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
        }
        write_bias(0);
        if (HEADS)
            for (int i = ctid; i < 6 * C; i += CTHR) reinterpret_cast<float*>(lds + HW_OFF)[i] = hd.w[i];
        int pp = -1, pb = 0;                            // the previous step (pair, block); pp < 0: none
        for (int p = 0; p < pairs; ++p)
            for (int b = 0; b < NB; ++b)
                for (int s = 0; s < 2; ++s) {
                    __syncthreads();                                    // A: X[s] holds this step's input
                    __syncthreads();                                    // B: the previous step's result is staged in X[so]
                    const int so = 1 - s;
                    if (pp >= 0) {
                        int board = -1;
                        if (pb == NB - 1) {                              // the chain's exit: to HBM
                            bool was_real;
                            board = board_of(pp, so, was_real);
                            if (!was_real) board = -1;
                        }
                        retire(so, pb, board);
                    }
                    // what X[so] needs next: the first board of slot B (very first step), or the next pair's board once the
                    // previous step was the chain's last block for it
                    if (pp < 0) fill_x(board_of(0, 1, real), so);
                    else if (pb == NB - 1 && pp + 1 < pairs) fill_x(board_of(pp + 1, so, real), so);
                    if (s == 1) write_bias(b + 1 < NB ? b + 1 : 0);     // the next block's bias (its buffer is read no more)
                    pp = p; pb = b;
                }
        __syncthreads();                                                // E1: the last step's result is staged in X[B]
        {
            bool was_real;
            const int board = board_of(pairs - 1, 1, was_real);
            retire(1, NB - 1, was_real ? board : -1);
        }
        return;
    }

    // ---- matrix waves ----
    const int kb = lane >> 5, ln = lane & 31;
    f32x16 acc[NT], prev[NT];
    tw::Shadow shd{lds, prev, wave, kb, ln, 0};
    bool have_prev = false;
    int prev_slot = 0;
    unsigned char* Y = lds + ROW_Y * RB;
    for (int p = 0; p < pairs; ++p)
        for (int b = 0; b < NB; ++b) {
            const c8k::Filter flt1 = c8k::make_filter(ch.w1[b], wave, lane), flt2 = c8k::make_filter(ch.w2[b], wave, lane);
            // the exponents of the images the two convolutions read (c8 filters carry 0: the c8 image has none)
            const int k_x = __builtin_amdgcn_readfirstlane(pack_ints(ch.w1[b])[2]);
            const int k_y = __builtin_amdgcn_readfirstlane(pack_ints(ch.w2[b])[2]);
            const int fx = fx_of(b), fy = fy_of(b);
            const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF) + (b & 1) * 2 * C;
            const float* bias2 = bias1 + C;
            for (int s = 0; s < 2; ++s) {
                unsigned char* X = lds + row_of(s) * RB;
                __syncthreads();                                       // A
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(bias1 + wave * 32 + g * 8 + kb * 4);
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        acc[q][g * 4 + 0] = bv.x; acc[q][g * 4 + 1] = bv.y; acc[q][g * 4 + 2] = bv.z; acc[q][g * 4 + 3] = bv.w;
                    }
                }
                shd.base = row_of(prev_slot) * RB;
                const c8k::Image imx{row_of(s), ROW_Z, PSTR};
                const int sx_lo = 127 + k_x - cf8::X_LO_SHIFT, sx = 127 + k_x;
                __builtin_amdgcn_s_setprio(3);
                if (fx == F_C6) {
                    if (have_prev) c8k::kloop<NT, tw::Shadow&, 0, false, 128, 1>(lds, imx, flt1, lane, acc, sx_lo, sx, shd);
                    else c8k::kloop<NT, c8k::NoShadow, 0, false, 128, 1>(lds, imx, flt1, lane, acc, sx_lo, sx);
                } else {
                    if (have_prev) c8k::kloop<NT, tw::Shadow&, 0, false, 128, 0>(lds, imx, flt1, lane, acc, sx_lo, sx, shd);
                    else c8k::kloop<NT, c8k::NoShadow, 0, false, 128, 0>(lds, imx, flt1, lane, acc, sx_lo, sx);
                }
                __builtin_amdgcn_s_setprio(0);
                int ln2 = ln, kb2 = kb;
                asm volatile("" : "+v"(ln2), "+v"(kb2));
                // ---- epilogue 1, first half: relu(acc) -> the operand triple of format fy -> Y (acc keeps relu(acc))
                if (fy == F_C6) {
                    // (k_resblock_c8's c6 path) the f16 quads go out per lane; the two bf6 pieces of a pixel's 32 channels need
                    // the other lane half's 16 values: tiles 0 and 1 trade halves, tile 2 trades with itself
                    const float s_hi = __builtin_ldexpf(1.0f, k_y), s_lo = __builtin_ldexpf(1.0f, k_y - cf8::X_LO_SHIFT);
                    f32x16 lo[NT];
#pragma unroll
                    for (int q3 = 0; q3 < NT; ++q3) {
                        const int q = q3 * 32 + ln2;
                        const int row = q < 90 ? q : 89;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = wave * 32 + g * 8 + kb2 * 4;
                            const int off = row * RB + (((c >> 3) ^ (row & 15)) << 4) + (c & 7) * 2;
                            Quad<_Float16> hq;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float r = acc[q3][g * 4 + i] > 0.0f ? acc[q3][g * 4 + i] : 0.0f;
                                hq.e[i] = (_Float16)r;
                                acc[q3][g * 4 + i] = r;
                                lo[q3][g * 4 + i] = r - (float)hq.e[i];
                            }
                            if (q < 90) *reinterpret_cast<Quad<_Float16>*>(Y + off) = hq;
                        }
                    }
#pragma unroll
                    for (int pq = 0; pq < 2; ++pq) {               // pair (0, 1), then tile 2 with itself
                        const int pa = pq == 0 ? 0 : 2, pb2 = pq == 0 ? 1 : 2;
                        f32x16 av, bv, al, bl;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[pa][r]), __float_as_uint(acc[pb2][r]), false, false);
                            const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[pa][r]), __float_as_uint(lo[pb2][r]), false, false);
                            av[r] = __uint_as_float(sv[0]); bv[r] = __uint_as_float(sv[1]);
                            al[r] = __uint_as_float(sl[0]); bl[r] = __uint_as_float(sl[1]);
                        }
                        const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
                        const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
                        const int q = pq == 0 ? kb2 * 32 + ln2 : 64 + ln2;
                        if (pq == 0 || (kb2 == 0 && q < 90)) {
                            unsigned char* P1 = Y + PSTR;
                            *reinterpret_cast<uint4*>(P1 + c6_lds_off(q, c6_chunk(0, wave))) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                            *reinterpret_cast<uint2*>(P1 + c6_lds_off(q, c6_chunk(0, wave) + 1)) = make_uint2(pl[4], pl[5]);
                            *reinterpret_cast<uint4*>(P1 + c6_lds_off(q, c6_chunk(1, wave))) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                            *reinterpret_cast<uint2*>(P1 + c6_lds_off(q, c6_chunk(1, wave) + 1)) = make_uint2(pv[4], pv[5]);
                        }
                    }
                } else {
                    // (k_resblock_c8's c8 path) f16 quad, e4m3 lo word, e4m3 value word per (tile, channel group)
#pragma unroll
                    for (int q3 = 0; q3 < NT; ++q3) {
                        const int q = q3 * 32 + ln2;
                        const int row = q < 90 ? q : 89;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = wave * 32 + g * 8 + kb2 * 4;
                            const int off = row * RB + (((c >> 3) ^ (row & 15)) << 4) + (c & 7) * 2;
                            const int off_lo = PSTR + row * RB + (c & 15) + (((c >> 4) ^ (row & 15)) << 4);
                            const int off_hi = PSTR + row * RB + (c & 15) + (((8 + (c >> 4)) ^ (row & 15)) << 4);
                            float r[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] = acc[q3][g * 4 + i] > 0.0f ? acc[q3][g * 4 + i] : 0.0f;
                            const cf8::Split4 o = cf8::split4(r);
                            if (q < 90) {
                                *reinterpret_cast<Quad<_Float16>*>(Y + off) = o.hi;
                                *reinterpret_cast<uint32_t*>(Y + off_lo) = o.l8;
                                *reinterpret_cast<uint32_t*>(Y + off_hi) = o.h8;
                            }
                        }
                    }
                }
                // ---- epilogue 1, second half: the accumulators restart at b2 + skip, this lane's own elements of X (format fx)
                if (fx == F_C6) {
                    const float s_skip = __builtin_ldexpf(1.0f, k_x - cf8::X_LO_SHIFT);
#pragma unroll
                    for (int q3 = 0; q3 < NT; ++q3) {
                        const int q = q3 * 32 + ln2;
                        const int row = q < 90 ? q : 89;
                        const uint4 hd4 = *reinterpret_cast<const uint4*>(X + PSTR + c6_lds_off(row, c6_chunk(0, wave)));
                        const uint2 tl2 = *reinterpret_cast<const uint2*>(X + PSTR + c6_lds_off(row, c6_chunk(0, wave) + 1));
                        // the upper lane half wants the odd elements: it shifts the piece down by one element (6 bits), so that
                        // every lane reads element 2 r for its register r
                        const uint32_t wv[7] = {hd4.x, hd4.y, hd4.z, hd4.w, tl2.x, tl2.y, 0u};
                        const uint32_t sh6 = (uint32_t)kb2 * 6u;
                        u32x6 pc;
#pragma unroll
                        for (int w = 0; w < 6; ++w) pc[w] = __builtin_amdgcn_alignbit(wv[w + 1], wv[w], sh6);
                        const f32x32 xl = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(pc, s_skip);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = wave * 32 + g * 8 + kb2 * 4;
                            const int off = row * RB + (((c >> 3) ^ (row & 15)) << 4) + (c & 7) * 2;
                            const float4 bv4 = *reinterpret_cast<const float4*>(bias2 + c);
                            float vv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
                            const Quad<_Float16> xq = *reinterpret_cast<const Quad<_Float16>*>(X + off);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r = g * 4 + i;
                                vv[i] += (float)xq.e[i] + xl[2 * r];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[q3][g * 4 + i] = vv[i];
                        }
                    }
                } else {
#pragma unroll
                    for (int q3 = 0; q3 < NT; ++q3) {
                        const int q = q3 * 32 + ln2;
                        const int row = q < 90 ? q : 89;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = wave * 32 + g * 8 + kb2 * 4;
                            const int off = row * RB + (((c >> 3) ^ (row & 15)) << 4) + (c & 7) * 2;
                            const int off_lo = PSTR + row * RB + (c & 15) + (((c >> 4) ^ (row & 15)) << 4);
                            const float4 bv4 = *reinterpret_cast<const float4*>(bias2 + c);
                            float vv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
                            cf8::add_pair4(vv, *reinterpret_cast<const Quad<_Float16>*>(X + off), *reinterpret_cast<const uint32_t*>(X + off_lo));
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[q3][g * 4 + i] = vv[i];
                        }
                    }
                }
                __syncthreads();                                       // B: Y complete; the previous step's staging complete
                const c8k::Image imy{ROW_Y, ROW_Z, PSTR};
                const int sy_lo = 127 + k_y - cf8::X_LO_SHIFT, sy = 127 + k_y;
                __builtin_amdgcn_s_setprio(3);
                if (fy == F_C6) c8k::kloop<NT, c8k::NoShadow, 0, false, 128, 1>(lds, imy, flt2, lane, acc, sy_lo, sy);
                else c8k::kloop<NT, c8k::NoShadow, 0, false, 128, 0>(lds, imy, flt2, lane, acc, sy_lo, sy);
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int q = 0; q < NT; ++q) prev[q] = acc[q];
                have_prev = true;
                prev_slot = s;
            }
        }
    // the last step's second epilogue, not overlapped: into its own image (slot B), dead since that step's barrier B
    shd.base = row_of(1) * RB;
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) shd.unit(prev[q], q * 32 + ln, g);
    __syncthreads();                                                    // E1
}

// ---- kernel: a chain of PAIR blocks ((hi, lo) fp16 / bf16 operands: f16x3 / bf16x3) ------------------------------------------------
// k_resblock_pipe's step -- K loop 1 on X[s] with the previous step's second epilogue (+ b2 + skip, ReLU, re-split) written IN
// PLACE over the other slot's skip operand in its shadow, epilogue 1 -> Y, K loop 2 -- leaves a block's result in the operand
// layout in the image the same slot's next block reads: chained over the blocks of a pair of boards nothing has to be converted
// or moved between blocks.  The copy waves fill a slot at the chain's entry and drain it (or apply the head convolutions to
// hi + lo) at its exit.  K loops, epilogues and their order are pipe_kloop's / PipeShadow's: bit-identical to n_blocks launches
// of k_resblock_pipe.
template <typename E, bool HEADS>
__global__ __launch_bounds__(512, 1) void k_tower_pairs(
    const E* __restrict__ xh, const E* __restrict__ xl, tw::Chain ch, E* __restrict__ yh, E* __restrict__ yl, int n_boards,
    const int32_t* __restrict__ n_dev, HeadArgs hd)
{
    using namespace tw;
    constexpr int NT = 3, CTHR = 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HEADS ? LDS_BYTES_HEADS : LDS_BYTES];
    if (n_dev) {
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stride = gridDim.x, t0 = blockIdx.x;
    if (t0 >= n_boards) return;
    const int NB = ch.n;
    const int mine = (n_boards - t0 + stride - 1) / stride;
    const int pairs = (mine + 1) / 2;
    auto board_of = [&](int p, int s, bool& real) __attribute__((always_inline)) {
        const int k = 2 * p + s;
        real = k < mine;
        return t0 + (real ? k : mine - 1) * stride;
    };
    auto row_of = [](int s) { return s ? ROW_XB : ROW_XA; };

    if (wave >= 4) {                                    // ---- copy waves ----
        const int ctid = tid - 256;
        // ONE mapping for every pass (k_tower's): item i = it * 256 + ctid < 360 is (pixel row i >> 2, 32-channel block i & 3)
        // = chunks 4 blk .. 4 blk + 3 of the row in both parts.  A thread only ever touches its own 128 bytes of a slot, so
        // exit (read) -> refill (write) needs no synchronisation among the copy waves.
        u32x4 v[2][8];                                  // [it][part * 4 + k]  (native vectors: see k_tower's fill_x)
        // (v[][] is defined on every path -- lanes without an item load a clamped one: see k_tower's fill_x)
        auto fetch = [&](int board) __attribute__((always_inline)) {
            const u32x4* sh = reinterpret_cast<const u32x4*>(xh + (size_t)board * 90 * C);
            const u32x4* sl = reinterpret_cast<const u32x4*>(xl + (size_t)board * 90 * C);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i0 = it * CTHR + ctid, i = i0 < 90 * 4 ? i0 : 90 * 4 - 1;
                const int qq = i >> 2, blk = i & 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[it][k] = sh[qq * 16 + blk * 4 + k];
                    v[it][4 + k] = sl[qq * 16 + blk * 4 + k];
                }
            }
        };
        auto put = [&](int s) __attribute__((always_inline)) {
            unsigned char* X = lds + row_of(s) * RB;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i < 90 * 4) {
                    const int qq = i >> 2, blk = i & 3;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int off = qq * RB + (((blk * 4 + k) ^ (qq & 15)) << 4);
                        *reinterpret_cast<u32x4*>(X + off) = v[it][k];
                        *reinterpret_cast<u32x4*>(X + PSTR + off) = v[it][4 + k];
                    }
                }
            }
        };
        // the result of `board` sits in X[s] (operand layout): to HBM; then the same chunks take the prefetched board
        auto drain = [&](int s, int board, bool refill) __attribute__((always_inline)) {
            unsigned char* X = lds + row_of(s) * RB;
            u32x4* dh = reinterpret_cast<u32x4*>(yh + (size_t)board * 90 * C);
            u32x4* dl = reinterpret_cast<u32x4*>(yl + (size_t)board * 90 * C);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i < 90 * 4) {
                    const int qq = i >> 2, blk = i & 3;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int off = qq * RB + (((blk * 4 + k) ^ (qq & 15)) << 4);
                        const u32x4 oh = *reinterpret_cast<const u32x4*>(X + off);
                        const u32x4 ol = *reinterpret_cast<const u32x4*>(X + PSTR + off);
                        if (refill) {
                            *reinterpret_cast<u32x4*>(X + off) = v[it][k];
                            *reinterpret_cast<u32x4*>(X + PSTR + off) = v[it][4 + k];
                        }
                        dh[qq * 16 + blk * 4 + k] = oh;
                        dl[qq * 16 + blk * 4 + k] = ol;
                    }
                }
            }
        };
        // HEADS exit: the six head features of every pixel of `board` from the value hi + lo of X[s] (k_tower's item mapping:
        // a thread = a pixel's 32-channel block, the pixel's four lanes add their partial dot products)
        auto heads_exit = [&](int s, int board) __attribute__((always_inline)) {
            const unsigned char* X = lds + row_of(s) * RB;
            const float* hwl = reinterpret_cast<const float*>(lds + HW_OFF);
            struct alignas(16) E8 { E e[8]; };
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = it * CTHR + ctid;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, blk = i & 3;
                float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int off = qq * RB + (((blk * 4 + k) ^ (qq & 15)) << 4);
                    const E8 h = __builtin_bit_cast(E8, *reinterpret_cast<const uint4*>(X + off));
                    const E8 l = __builtin_bit_cast(E8, *reinterpret_cast<const uint4*>(X + PSTR + off));
#pragma unroll
                    for (int o = 0; o < 6; ++o) {
                        const float4 w0 = *reinterpret_cast<const float4*>(hwl + o * C + blk * 32 + 8 * k);
                        const float4 w1 = *reinterpret_cast<const float4*>(hwl + o * C + blk * 32 + 8 * k + 4);
                        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[o] += ((float)h.e[j] + (float)l.e[j]) * w[j];
                    }
                }
#pragma unroll
                for (int o = 0; o < 6; ++o) {
                    a[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[o]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    a[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[o]), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                }
                if (board < 0) continue;
#pragma unroll
                for (int o = 0; o < 6; ++o)
                    if ((o & 3) == blk) {
                        float hv = a[o] + hd.b[o];
                        hv = hv > 0.0f ? hv : 0.0f;
                        if (o < hd.n_pol) hd.pol[(size_t)board * (hd.n_pol * 90) + o * 90 + qq] = hv;
                        else hd.val[(size_t)board * ((6 - hd.n_pol) * 90) + (o - hd.n_pol) * 90 + qq] = hv;
                    }
            }
        };
        // (bias buffer = parity of the RUNNING block count g = pair * NB + block, not of the block index: the second epilogue of
        //  a pair's last step reads the last block's b2 while the next pair's first step already reads block 0's b1 -- with an
        //  odd number of blocks both would sit in buffer 0)
        auto write_bias = [&](int b, int g) __attribute__((always_inline)) {
            if (ctid < C) {
                float* dst = reinterpret_cast<float*>(lds + BIAS_OFF) + (g & 1) * 2 * C;
                dst[ctid] = ch.b1[b][ctid];
                dst[C + ctid] = ch.b2[b][ctid];
            }
        };
        bool real;
        fetch(board_of(0, 0, real));
        put(0);
        for (int i = ctid; i < 16 * 16; i += CTHR) {
            *reinterpret_cast<uint4*>(lds + ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + PSTR + ROW_Z * RB + i * 16) = make_uint4(0, 0, 0, 0);
        }
        write_bias(0, 0);
        if (HEADS)
            for (int i = ctid; i < 6 * C; i += CTHR) reinterpret_cast<float*>(lds + HW_OFF)[i] = hd.w[i];
        int pp = -1, pb = 0;
        for (int p = 0; p < pairs; ++p)
            for (int b = 0; b < NB; ++b)
                for (int s = 0; s < 2; ++s) {
                    __syncthreads();                                    // A: X[s] holds this step's input
                    const int so = 1 - s;
                    bool fill = false;
                    if (pp < 0) { fill = true; fetch(board_of(0, 1, real)); }
                    else if (pb == NB - 1 && pp + 1 < pairs) { fill = true; fetch(board_of(pp + 1, so, real)); }
                    __syncthreads();                                    // B: the previous step's result is complete in X[so]
                    if (pp >= 0 && pb == NB - 1) {                       // the chain's exit for slot so
                        bool was_real;
                        const int board = board_of(pp, so, was_real);
                        if (HEADS) {
                            heads_exit(so, was_real ? board : -1);
                            if (fill) put(so);                           // (a thread overwrites only what it has read itself)
                        } else if (was_real) {
                            drain(so, board, fill);
                        } else if (fill) {
                            put(so);
                        }
                    } else if (fill) {
                        put(so);
                    }
                    if (s == 1) write_bias(b + 1 < NB ? b + 1 : 0, p * NB + b + 1);
                    pp = p; pb = b;
                }
        __syncthreads();                                                // E1: K loop 2 of the last step is done
        __syncthreads();                                                // E2: its second epilogue is written in place in X[B]
        {
            bool was_real;
            const int board = board_of(pairs - 1, 1, was_real);
            if (HEADS) heads_exit(1, was_real ? board : -1);
            else if (was_real) drain(1, board, false);
        }
        return;
    }

    // ---- matrix waves ----
    const int kb = lane >> 5, ln = lane & 31;
    f32x16 acc[NT], prev[NT];
    PipeShadow<E> shd;
    shd.lds = lds; shd.wave = wave; shd.kb = kb; shd.ln = ln;
    bool have_prev = false;
    int prev_slot = 0, prev_g = 0;
    for (int p = 0; p < pairs; ++p)
        for (int b = 0; b < NB; ++b) {
            const uint4* wq1 = reinterpret_cast<const uint4*>(ch.w1[b]) + wave * 64 + lane;
            const uint4* wq2 = reinterpret_cast<const uint4*>(ch.w2[b]) + wave * 64 + lane;
            const int g = p * NB + b;                                  // running block count: its parity picks the bias buffer
            const int bias1_off = BIAS_OFF + (g & 1) * 2 * C * 4;
            for (int s = 0; s < 2; ++s) {
                __syncthreads();                                       // A
                shd.prev_row_base = row_of(prev_slot);
                shd.bias2_off = BIAS_OFF + ((prev_g & 1) * 2 * C + C) * 4;
                __builtin_amdgcn_s_setprio(3);
                if (have_prev) pipe_kloop<E, true>(lds, row_of(s), wq1, lane, acc, prev, shd);
                else pipe_kloop<E, false>(lds, row_of(s), wq1, lane, acc, prev, shd);
                __builtin_amdgcn_s_setprio(0);
                int ln2 = ln, kb2 = kb;
                asm volatile("" : "+v"(ln2), "+v"(kb2));
                // epilogue 1 (k_resblock_pipe's): relu(acc + b1) -> (hi, lo) -> Y
#pragma unroll
                for (int q3 = 0; q3 < NT; ++q3) {
                    const int q = q3 * 32 + ln2;
                    if (q < 90) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = wave * 32 + g * 8 + kb2 * 4;
                            const float4 bv = *reinterpret_cast<const float4*>(lds + bias1_off + c * 4);
                            const float vv[4] = {acc[q3][g * 4 + 0] + bv.x, acc[q3][g * 4 + 1] + bv.y, acc[q3][g * 4 + 2] + bv.z,
                                                 acc[q3][g * 4 + 3] + bv.w};
                            Quad<E> hi, lo;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                                hi.e[i] = (E)r;
                                lo.e[i] = (E)(r - (float)hi.e[i]);
                            }
                            const int off = (ROW_Y + q) * RB + (((c >> 3) ^ (q & 15)) << 4) + (c & 7) * 2;
                            *reinterpret_cast<Quad<E>*>(lds + off) = hi;
                            *reinterpret_cast<Quad<E>*>(lds + PSTR + off) = lo;
                        }
                    }
                }
                __syncthreads();                                       // B: Y complete (and the previous step's result, written during K loop 1)
                __builtin_amdgcn_s_setprio(3);
                pipe_kloop<E, false>(lds, ROW_Y, wq2, lane, acc, prev, shd);
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int q = 0; q < NT; ++q) prev[q] = acc[q];
                have_prev = true;
                prev_slot = s;
                prev_g = g;
            }
        }
    __syncthreads();                                                    // E1
    shd.prev_row_base = row_of(1);                                      // the last step's second epilogue, not overlapped
    shd.bias2_off = BIAS_OFF + ((prev_g & 1) * 2 * C + C) * 4;
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) shd.whole(prev[q], q, g);
    __syncthreads();                                                    // E2
}

}  // namespace

// ---- entry points ---------------------------------------------------------------------------------------------------------------
static int tower_cu_count()
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

// One chain of staged blocks.  fmt_x / fmt_y: per block, CZ_IMG_C8 or CZ_IMG_C6 (NULL: all CZ_IMG_C6); exit_fmt: CZ_IMG_C8 /
// CZ_IMG_C6 / CZ_IMG_PAIR, or CZ_EXIT_HEADS.
// ---- kernel: the chain of PAIR blocks on four matrix waves (round 6) -----------------------------------------------------------
// k_tower_pairs' arithmetic (three MFMAs per product: w_hi x_hi, w_lo x_hi, w_hi x_lo in that order per K-step; epilogue 1
// relu(acc + b1) -> (hi, lo); epilogue 2 ((acc + b2) + skip_hi) + skip_lo, relu, (hi, lo)) in k_resblock_ip4_c8's shape: a
// pair of boards per workgroup with ONE image each (A = rows [0, 90), B = rows [90, 180)), four matrix waves of 512 registers, no
// copy waves -- wave w takes board w >> 1 and channel tiles 2 (w & 1), + 1, so a pixel fragment read from LDS feeds two MFMAs
// per pass.  After K loop 1 (barrier: both waves of a board have read its image) a lane moves its skip elements (hi and lo
// quads of its own channels: 96 registers) out of the image and writes the intermediate activation over them; epilogue 2 writes
// the block's result to the same bytes.  Exits: the (hi, lo) pair to HBM, or the head features from hi + lo of the result as
// k_tower_pairs forms them, item for item.  Bit-identical to k_tower_pairs.
namespace tw4 {
constexpr int MAX_BLOCKS = 12;
struct Chain {
    const void* w1[MAX_BLOCKS];
    const void* w2[MAX_BLOCKS];
    const float* b1[MAX_BLOCKS];
    const float* b2[MAX_BLOCKS];
    int n;
};
}  // namespace tw4

template <int CH> struct Tp4 {                  // geometry for CH filters (128: 256-byte rows; 192: 384-byte rows, chunks swizzled in groups of 8)
    static constexpr int C = CH, RB = 2 * CH, CPR = CH / 8, NT = 3, CT = CH / 32, CTW = CT / 2, NTHR = 256, KK = CH / 16;
    static constexpr bool POW2 = (RB & (RB - 1)) == 0;
    static constexpr int SWZ = POW2 ? 15 : 7;
    static constexpr int ROW_Z = 192, PSTR = (ROW_Z + 16) * RB;
    static constexpr int BIAS_OFF = 2 * PSTR, HW_OFF = BIAS_OFF + 2 * 2 * CH * 4;
    static constexpr int LDS_BYTES = HW_OFF + (CH == 128 ? 6 * CH * 4 : 0);      // (192 filters: no room for the head filters, no heads exit)
    static constexpr int W_STEP = CT * 64, W_PART = (9 * KK + W_PAD_STEPS) * W_STEP, W_RING = CH == 128 ? 4 : 3;
    static_assert(CT == 2 * CTW && KK % W_RING == 0 && KK % 2 == 0 && LDS_BYTES <= 160 * 1024, "two waves of CTW channel tiles per board");
    // byte offset of K-step kk relative to a tap's row offset (the lane's kb ^ row bits folded in)
    static __device__ __forceinline__ int kstep(int pre, int kk) { return POW2 ? pre ^ (kk << 5) : (pre ^ ((kk & 3) << 5)) + ((kk >> 2) << 7); }
    // 16-byte chunk `chunk` of pixel row `key` of board `bd` (inside a part): the swizzle key is the board-relative row
    static __device__ __forceinline__ int choff(int bd, int key, int chunk)
    {
        return (bd * 90 + key) * RB + ((chunk & ~SWZ) << 4) + (((chunk ^ key) & SWZ) << 4);
    }
};

template <typename E, int CH>
__device__ __forceinline__ void pairs_kloop_ctw(const unsigned char* lds, int row_base, const uint4* wq, int lane, f32x16* acc)
{
    typedef Tp4<CH> G;
    constexpr int RB = G::RB, NT = G::NT, CTW = G::CTW, KK = G::KK, ROW_Z = G::ROW_Z, PSTR = G::PSTR, W_STEP = G::W_STEP,
                  W_PART = G::W_PART, W_RING = G::W_RING;
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
        return row * RB + (((kb ^ nominal) & G::SWZ) << 4);   // swizzle key = the board-relative row
    };
    V8 wf[W_RING][CTW][2];
    V8 px[2][NT][2];
#pragma unroll
    for (int p = 0; p < CTW * NT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    auto load_w = [&](int step, int c, int part) {
        return __builtin_bit_cast(V8, wq[(size_t)part * W_PART + (size_t)step * W_STEP + c * 64]);
    };
    auto load_px = [&](int off, int part) {
        return __builtin_bit_cast(V8, *reinterpret_cast<const c8k::u32x4*>(lds + part * PSTR + off));
    };
#pragma unroll
    for (int p = 0; p < NT; ++p) pre[p] = tap_row(-1, -1, p);
#pragma unroll
    for (int s = 0; s < W_RING - 1; ++s)
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int part = 0; part < 2; ++part) wf[s][c][part] = load_w(s, c, part);
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int p = 0; p < NT; ++p) px[0][p][part] = load_px(pre[p], part);
    constexpr int NM = 3 * NT * CTW, NL = NT * 2, NW = CTW * 2;
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int tap = 3 * j + tt;
            const int ndy = tt < 2 ? j - 1 : (j < 2 ? j : 1), ndx = tt < 2 ? tt : -1;   // the NEXT tap (last: a valid one, unused)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const int step = tap * KK + kk;
                const int* rows = kk + 1 < KK ? pre : pre_n;
                const int kn = (kk + 1) % KK;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const int pass = i / (NT * CTW), p = (i % (NT * CTW)) / CTW, c = i % CTW;
                    acc[c * NT + p] = Mfma<E>::mma(wf[kk % W_RING][c][pass == 1 ? 1 : 0], px[kk & 1][p][pass == 2 ? 1 : 0], acc[c * NT + p]);
                    if (i < NL) px[(kk + 1) & 1][i % NT][i / NT] = load_px(G::kstep(rows[i % NT], kn), i / NT);
                    if (i == NM - 1 - NW && kk < NT) pre_n[kk] = tap_row(ndy, ndx, kk);
                    if (i >= NM - NW) {
                        const int idx = i - (NM - NW);
                        wf[(kk + W_RING - 1) % W_RING][idx / 2][idx % 2] = load_w(step + W_RING - 1, idx / 2, idx % 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) pre[p] = pre_n[p];
        }
    }
}

template <typename E, int CH>
__global__ __launch_bounds__(256, 1) void k_tower_pairs4(
    const E* __restrict__ xh, const E* __restrict__ xl, tw4::Chain ch, E* __restrict__ yh, E* __restrict__ yl, int n_boards,
    const int32_t* __restrict__ n_dev, HeadArgs hd, int heads, float* __restrict__ yf_last)
{
    typedef Tp4<CH> G;
    constexpr int C = G::C, RB = G::RB, CPR = G::CPR, NT = G::NT, CTW = G::CTW, NTHR = G::NTHR, ROW_Z = G::ROW_Z, PSTR = G::PSTR,
                  BIAS_OFF = G::BIAS_OFF, HW_OFF = G::HW_OFF;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS_BYTES];
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
    constexpr int CHUNKS = 180 * CPR, LITER = (CHUNKS + NTHR - 1) / NTHR;
    auto choff = [&](int bd, int key, int chunk) { return G::choff(bd, key, chunk); };
    auto chunk_off = [&](int i) {
        const int row = i / CPR, c = i - row * CPR;
        return choff(row >= 90 ? 1 : 0, row >= 90 ? row - 90 : row, c);
    };
    auto fill = [&](int pr) __attribute__((always_inline)) {    // HBM -> images (a missing second board: zeros)
        const int have = (n_boards - 2 * pr < 2 ? n_boards - 2 * pr : 2) * 90 * CPR;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const u4* src = reinterpret_cast<const u4*>((part ? xl : xh) + (size_t)2 * pr * 90 * C);
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
                    if (it0 + j < LITER && i < CHUNKS) *reinterpret_cast<u4*>(lds + part * PSTR + chunk_off(i)) = v[j];
                }
            }
        }
    };
    auto drain = [&](int pr) __attribute__((always_inline)) {
        const int have = (n_boards - 2 * pr < 2 ? n_boards - 2 * pr : 2) * 90 * CPR;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            u4* dst = reinterpret_cast<u4*>((part ? yl : yh) + (size_t)2 * pr * 90 * C);
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
        if (tid < C) {
            dst[tid] = ch.b1[blk][tid];
            dst[C + tid] = ch.b2[blk][tid];
        }
        static_assert(C <= NTHR, "one thread per channel");
    };
    for (int i = tid; i < 16 * CPR; i += NTHR) {                // the shared zero rows, both parts
        *reinterpret_cast<u4*>(lds + ROW_Z * RB + i * 16) = u4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u4*>(lds + PSTR + ROW_Z * RB + i * 16) = u4{0u, 0u, 0u, 0u};
    }
    fill(t);
    write_bias(0);
    write_bias(1);
    if (C == 128 && heads)
        for (int i = tid; i < 6 * C; i += NTHR) reinterpret_cast<float*>(lds + HW_OFF)[i] = hd.w[i];

    const int kb = lane >> 5, ln = lane & 31;
    const int bd = wave >> 1, tile0 = CTW * (wave & 1);
    int g = 0;
    for (;;) {
        __syncthreads();                                        // A: the images hold pair t, the bias buffers are written
        for (int blk = 0; blk < NB; ++blk, ++g) {
            const uint4* wq1 = reinterpret_cast<const uint4*>(ch.w1[blk]) + tile0 * 64 + lane;
            const uint4* wq2 = reinterpret_cast<const uint4*>(ch.w2[blk]) + tile0 * 64 + lane;
            const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF) + (g & 1) * 2 * C;
            const float* bias2 = bias1 + C;
            f32x16 acc[CTW * NT];
            c8k::u32x2 skh[CTW * NT][4], skl[CTW * NT][4];      // the skip operand's (hi, lo) quads, packed
            __builtin_amdgcn_s_setprio(3);
            pairs_kloop_ctw<E, CH>(lds, bd * 90, wq1, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();                                    // K1: both waves of a board have read its image
            int ln2 = ln, kb2 = kb;
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 1, in place: skip <- image, image <- relu(acc + b1) as (hi, lo)
#pragma unroll
            for (int cp = 0; cp < CTW * NT; ++cp) {
                const int c = cp / NT, p = cp % NT;
                const int q = p * 32 + ln2;
                const int key = q < 90 ? q : 89;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int chn = (tile0 + c) * 32 + gg * 8 + kb2 * 4;
                    const int off = choff(bd, key, chn >> 3) + (chn & 7) * 2;
                    skh[cp][gg] = *reinterpret_cast<const c8k::u32x2*>(lds + off);
                    skl[cp][gg] = *reinterpret_cast<const c8k::u32x2*>(lds + PSTR + off);
                    const float4 bv = *reinterpret_cast<const float4*>(bias1 + chn);
                    const float vv[4] = {acc[cp][gg * 4 + 0] + bv.x, acc[cp][gg * 4 + 1] + bv.y, acc[cp][gg * 4 + 2] + bv.z,
                                         acc[cp][gg * 4 + 3] + bv.w};
                    Quad<E> hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = vv[i] > 0.0f ? vv[i] : 0.0f;
                        hi.e[i] = (E)r;
                        lo.e[i] = (E)(r - (float)hi.e[i]);
                    }
                    if (q < 90) {
                        *reinterpret_cast<Quad<E>*>(lds + off) = hi;
                        *reinterpret_cast<Quad<E>*>(lds + PSTR + off) = lo;
                    }
                }
            }
            __syncthreads();                                    // B: the images hold the intermediate activation; block g's b1 is consumed
            __builtin_amdgcn_s_setprio(3);
            pairs_kloop_ctw<E, CH>(lds, bd * 90, wq2, lane, acc);
            __builtin_amdgcn_s_setprio(0);
            __syncthreads();                                    // K2
            asm volatile("" : "+v"(ln2), "+v"(kb2));
            // epilogue 2, in place: image <- relu(((acc + b2) + skip_hi) + skip_lo) as (hi, lo)
#pragma unroll
            for (int cp = 0; cp < CTW * NT; ++cp) {
                const int c = cp / NT, p = cp % NT;
                const int q = p * 32 + ln2;
                if (q < 90) {
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        const int chn = (tile0 + c) * 32 + gg * 8 + kb2 * 4;
                        const int off = choff(bd, q, chn >> 3) + (chn & 7) * 2;
                        const float4 bv = *reinterpret_cast<const float4*>(bias2 + chn);
                        const Quad<E> sh = __builtin_bit_cast(Quad<E>, skh[cp][gg]), sl = __builtin_bit_cast(Quad<E>, skl[cp][gg]);
                        float v[4] = {acc[cp][gg * 4 + 0] + bv.x, acc[cp][gg * 4 + 1] + bv.y, acc[cp][gg * 4 + 2] + bv.z,
                                      acc[cp][gg * 4 + 3] + bv.w};
                        Quad<E> hi, lo;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] += (float)sh.e[i];
                            v[i] += (float)sl.e[i];
                            v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                            hi.e[i] = (E)v[i];
                            lo.e[i] = (E)(v[i] - (float)hi.e[i]);
                        }
                        if (yf_last && blk == NB - 1) {          // the tower's last block: its fp32 value for the head convolutions
                            const int board = 2 * t + bd;
                            if (board < n_boards)
                                *reinterpret_cast<float4*>(yf_last + ((size_t)board * 90 + q) * C + chn) = make_float4(v[0], v[1], v[2], v[3]);
                            continue;
                        }
                        *reinterpret_cast<Quad<E>*>(lds + off) = hi;
                        *reinterpret_cast<Quad<E>*>(lds + PSTR + off) = lo;
                    }
                }
            }
            __syncthreads();                                    // C: the block's result is in the images; its b2 is consumed
            if (NB > 1) write_bias(g + 2);                      // (into the buffer block g has just released)
        }
        if (C == 128 && heads) {
            // the head features of board 2 t + bd by its two waves, from hi + lo of the result (k_tower_pairs' heads_exit)
            const int board = 2 * t + bd;
            const float* hwl = reinterpret_cast<const float*>(lds + HW_OFF);
            struct alignas(16) E8 { E e[8]; };
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int i = it * 128 + (wave & 1) * 64 + lane;
                if (i >= 90 * 4) continue;
                const int qq = i >> 2, b32 = i & 3;
                float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int off = choff(bd, qq, b32 * 4 + k);
                    const E8 h = __builtin_bit_cast(E8, *reinterpret_cast<const u4*>(lds + off));
                    const E8 l = __builtin_bit_cast(E8, *reinterpret_cast<const u4*>(lds + PSTR + off));
#pragma unroll
                    for (int o = 0; o < 6; ++o) {
                        const float4 w0 = *reinterpret_cast<const float4*>(hwl + o * C + b32 * 32 + 8 * k);
                        const float4 w1 = *reinterpret_cast<const float4*>(hwl + o * C + b32 * 32 + 8 * k + 4);
                        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) a[o] += ((float)h.e[jj] + (float)l.e[jj]) * w[jj];
                    }
                }
#pragma unroll
                for (int o = 0; o < 6; ++o) {
                    a[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[o]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    a[o] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[o]), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                }
                if (board >= n_boards) continue;
#pragma unroll
                for (int o = 0; o < 6; ++o)
                    if ((o & 3) == b32) {
                        float hv = a[o] + hd.b[o];
                        hv = hv > 0.0f ? hv : 0.0f;
                        if (o < hd.n_pol) hd.pol[(size_t)board * (hd.n_pol * 90) + o * 90 + qq] = hv;
                        else hd.val[(size_t)board * ((6 - hd.n_pol) * 90) + (o - hd.n_pol) * 90 + qq] = hv;
                    }
            }
        } else if (!yf_last) {
            drain(t);
        }
        t += stride;
        if (t >= n_pairs) break;
        __syncthreads();                                        // (the exit has read the images)
        fill(t);
    }
}

// the pair chains' launches (cz_tower_pairs: 128 filters; cz_resblock_chain in csrc/xq_conv.hip: 192 filters, no heads exit)
extern "C" int czi_pairs4_launch(const void* x_hi, const void* x_lo, int n_blocks, const void* const* w1, const float* const* b1,
                                 const void* const* w2, const float* const* b2, void* y_hi, void* y_lo, const float* head_w,
                                 const float* head_b, float* pol, float* val, int n_pol, int n_boards, int channels, int dtype,
                                 int n_cu, const int32_t* n_dev, void* stream, float* y_f32)
{
    if (n_blocks > tw4::MAX_BLOCKS || (channels != 128 && channels != 192) || (head_w && channels != 128)) return CZ_ERR_ARG;
    tw4::Chain ch{};
    ch.n = n_blocks;
    for (int b = 0; b < n_blocks; ++b) { ch.w1[b] = w1[b]; ch.w2[b] = w2[b]; ch.b1[b] = b1[b]; ch.b2[b] = b2[b]; }
    const HeadArgs hd = head_w ? HeadArgs{head_w, head_b, pol, val, n_pol} : HeadArgs{};
    const int n_pairs = (n_boards + 1) / 2;
    const unsigned blocks = (unsigned)(n_pairs < n_cu ? n_pairs : n_cu);
    hipStream_t st = (hipStream_t)stream;
#define CZ_P4(E, CH) hipLaunchKernelGGL((k_tower_pairs4<E, CH>), dim3(blocks), dim3(256), 0, st, (const E*)x_hi, (const E*)x_lo, ch, \
                                        (E*)y_hi, (E*)y_lo, n_boards, n_dev, hd, head_w ? 1 : 0, y_f32)
    if (channels == 128) { if (dtype == CZ_F16) CZ_P4(_Float16, 128); else CZ_P4(__bf16, 128); }
    else { if (dtype == CZ_F16) CZ_P4(_Float16, 192); else CZ_P4(__bf16, 192); }
#undef CZ_P4
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

extern "C" int cz_tower(const void* x_hi, const void* x_img, int n_blocks, const void* const* w1_packed,
                        const float* const* bias1, const void* const* w2_packed, const float* const* bias2,
                        const int* fmt_x, const int* fmt_y, int exit_fmt, void* y_hi, void* y_img, const float* head_w,
                        const float* head_b, float* policy_feat, float* value_feat, int n_policy, int n_value, int n_boards,
                        const int32_t* n_dev, void* stream)
{
    const bool heads = exit_fmt == CZ_EXIT_HEADS;
    if (n_boards < 0 || !x_hi || !x_img || !w1_packed || !w2_packed || !bias1 || !bias2 || n_blocks < 1 ||
        n_blocks > tw::MAX_BLOCKS || (!heads && (!y_hi || !y_img)) ||
        (heads && (!head_w || !head_b || !policy_feat || !value_feat || n_policy < 1 || n_value < 1 || n_policy + n_value != 6)) ||
        (!heads && exit_fmt != CZ_IMG_C8 && exit_fmt != CZ_IMG_C6 && exit_fmt != CZ_IMG_PAIR)) {
        czi_set_error("cz_tower: bad argument (1 .. 8 blocks; exit CZ_IMG_C8 / CZ_IMG_C6 / CZ_IMG_PAIR with y_hi + y_img, or "
                      "CZ_EXIT_HEADS with n_policy + n_value == 6)");
        return CZ_ERR_ARG;
    }
    tw::Chain ch{};
    ch.n = n_blocks;
    const int fmt0 = fmt_x ? fmt_x[0] : CZ_IMG_C6;
    ch.exit_fmt = heads ? fmt0 : exit_fmt;
    for (int b = 0; b < n_blocks; ++b) {
        if (!w1_packed[b] || !w2_packed[b] || !bias1[b] || !bias2[b]) {
            czi_set_error("cz_tower: null block parameter");
            return CZ_ERR_ARG;
        }
        const int fx = fmt_x ? fmt_x[b] : CZ_IMG_C6, fy = fmt_y ? fmt_y[b] : CZ_IMG_C6;
        if ((fx != CZ_IMG_C8 && fx != CZ_IMG_C6) || fy != fx || fx != fmt0) {
            czi_set_error("cz_tower: one image format per chain, CZ_IMG_C8 or CZ_IMG_C6 (a hybrid tower is one chain per "
                          "arithmetic -- the exit of the first hands over; pair blocks: cz_tower_pairs)");
            return CZ_ERR_ARG;
        }
        ch.w1[b] = w1_packed[b]; ch.w2[b] = w2_packed[b]; ch.b1[b] = bias1[b]; ch.b2[b] = bias2[b];
        ch.fx[b] = (unsigned char)fx; ch.fy[b] = (unsigned char)fy;
    }
    if (!heads && ((fmt0 == CZ_IMG_C6 && exit_fmt == CZ_IMG_PAIR) || (fmt0 == CZ_IMG_C8 && exit_fmt == CZ_IMG_C6))) {
        czi_set_error("cz_tower: a c6 chain ends on a c6 or c8 image, a c8 chain on a c8 image or fp16 pairs");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = tower_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_tower: cannot query the device");
        return CZ_ERR_HIP;
    }
    const unsigned blocks = (unsigned)(n_boards < n_cu ? n_boards : n_cu);
    const HeadArgs hd = heads ? HeadArgs{head_w, head_b, policy_feat, value_feat, n_policy} : HeadArgs{};
    hipStream_t st = (hipStream_t)stream;
    // round 6: the chain on FOUR matrix waves of two channel tiles, a pair of boards with one image each and both epilogues in
    // place (k_resblock_ip4_c8<128>, csrc/xq_conv.hip; built for the 192-filter tower and 9 % faster per block than k_tower
    // here: a pixel fragment feeds two MFMAs, no copy waves).  Same exits, bit-identical.  CZ_TOWER4=0: k_tower.
    const char* t4 = getenv("CZ_TOWER4");
    if (!(t4 && t4[0] == '0')) {
        const int rc = czi_tower4_launch(x_hi, x_img, n_blocks, w1_packed, bias1, w2_packed, bias2, fmt0 == CZ_IMG_C6,
                                         heads ? 3 : (exit_fmt == CZ_IMG_PAIR ? 2 : 0), y_hi, y_img, head_w, head_b, policy_feat,
                                         value_feat, n_policy, n_boards, n_cu, n_dev, stream);
        if (rc != CZ_OK) czi_set_error("cz_tower: launch failed");
        return rc;
    }
#define CZ_TOWER_LAUNCH(H, U) hipLaunchKernelGGL((k_tower<H, U>), dim3(blocks), dim3(512), 0, st, (const _Float16*)x_hi, \
        (const unsigned char*)x_img, ch, (_Float16*)y_hi, (unsigned char*)y_img, n_boards, n_dev, hd)
    if (fmt0 == CZ_IMG_C6) {
        if (heads) CZ_TOWER_LAUNCH(true, CZ_IMG_C6); else CZ_TOWER_LAUNCH(false, CZ_IMG_C6);
    } else {
        if (heads) CZ_TOWER_LAUNCH(true, CZ_IMG_C8); else CZ_TOWER_LAUNCH(false, CZ_IMG_C8);
    }
#undef CZ_TOWER_LAUNCH
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_tower: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}

// (round 5's names: all blocks c6)
extern "C" int cz_tower_c6(const void* x_hi, const void* x_c6, int n_blocks, const void* const* w1_packed,
                           const float* const* bias1, const void* const* w2_packed, const float* const* bias2, void* y_hi,
                           void* y_c6, int n_boards, const int32_t* n_dev, void* stream)
{
    if (n_blocks < 2 || !y_hi || !y_c6) {
        czi_set_error("cz_tower_c6: bad argument (2 .. 8 blocks, c6 operand pairs in and out)");
        return CZ_ERR_ARG;
    }
    return cz_tower(x_hi, x_c6, n_blocks, w1_packed, bias1, w2_packed, bias2, nullptr, nullptr, CZ_IMG_C6, y_hi, y_c6, nullptr,
                    nullptr, nullptr, nullptr, 0, 0, n_boards, n_dev, stream);
}

extern "C" int cz_tower_c6_heads(const void* x_hi, const void* x_c6, int n_blocks, const void* const* w1_packed,
                                 const float* const* bias1, const void* const* w2_packed, const float* const* bias2,
                                 const float* head_w, const float* head_b, float* policy_feat, float* value_feat, int n_boards,
                                 int n_policy, int n_value, const int32_t* n_dev, void* stream)
{
    if (n_blocks < 2) {
        czi_set_error("cz_tower_c6_heads: bad argument (2 .. 8 blocks, c6 operand pair in, n_policy + n_value == 6)");
        return CZ_ERR_ARG;
    }
    return cz_tower(x_hi, x_c6, n_blocks, w1_packed, bias1, w2_packed, bias2, nullptr, nullptr, CZ_EXIT_HEADS, nullptr, nullptr,
                    head_w, head_b, policy_feat, value_feat, n_policy, n_value, n_boards, n_dev, stream);
}

// A chain of PAIR blocks ((hi, lo) operands of dtype CZ_F16 or CZ_BF16; cz_conv3x3_pack_weights filters with parts = 2):
// n_blocks launches of cz_resblock in one (bit-identical).  head_w != NULL: the chain ends on the tower's last block and writes
// the head features instead of y.
extern "C" int cz_tower_pairs(const void* x_hi, const void* x_lo, int n_blocks, const void* const* w1_packed,
                              const float* const* bias1, const void* const* w2_packed, const float* const* bias2, void* y_hi,
                              void* y_lo, const float* head_w, const float* head_b, float* policy_feat, float* value_feat,
                              int n_policy, int n_value, int n_boards, int dtype, const int32_t* n_dev, void* stream)
{
    const bool heads = head_w != nullptr;
    if (n_boards < 0 || !x_hi || !x_lo || !w1_packed || !w2_packed || !bias1 || !bias2 || n_blocks < 1 ||
        n_blocks > tw::MAX_BLOCKS || (dtype != CZ_F16 && dtype != CZ_BF16) || (!heads && (!y_hi || !y_lo)) ||
        (heads && (!head_b || !policy_feat || !value_feat || n_policy < 1 || n_value < 1 || n_policy + n_value != 6))) {
        czi_set_error("cz_tower_pairs: bad argument (1 .. 8 blocks of (hi, lo) f16 / bf16 operands; y_hi + y_lo, or the head "
                      "arguments with n_policy + n_value == 6)");
        return CZ_ERR_ARG;
    }
    tw::Chain ch{};
    ch.n = n_blocks;
    ch.exit_fmt = CZ_IMG_PAIR;
    for (int b = 0; b < n_blocks; ++b) {
        if (!w1_packed[b] || !w2_packed[b] || !bias1[b] || !bias2[b]) {
            czi_set_error("cz_tower_pairs: null block parameter");
            return CZ_ERR_ARG;
        }
        ch.w1[b] = w1_packed[b]; ch.w2[b] = w2_packed[b]; ch.b1[b] = bias1[b]; ch.b2[b] = bias2[b];
        ch.fx[b] = ch.fy[b] = (unsigned char)CZ_IMG_PAIR;
    }
    if (n_boards == 0) return CZ_OK;
    const int n_cu = tower_cu_count();
    if (n_cu < 0) {
        czi_set_error("cz_tower_pairs: cannot query the device");
        return CZ_ERR_HIP;
    }
    const unsigned blocks = (unsigned)(n_boards < n_cu ? n_boards : n_cu);
    const HeadArgs hd = heads ? HeadArgs{head_w, head_b, policy_feat, value_feat, n_policy} : HeadArgs{};
    hipStream_t st = (hipStream_t)stream;
    const char* t4 = getenv("CZ_TOWER4");                       // (round 6) the chain on four matrix waves; CZ_TOWER4=0: k_tower_pairs
    if (!(t4 && t4[0] == '0')) {
        const int rc = czi_pairs4_launch(x_hi, x_lo, n_blocks, w1_packed, bias1, w2_packed, bias2, y_hi, y_lo, head_w, head_b,
                                         policy_feat, value_feat, n_policy, n_boards, 128, dtype, n_cu, n_dev, stream, nullptr);
        if (rc != CZ_OK) czi_set_error("cz_tower_pairs: launch failed");
        return rc;
    }
#define CZ_PAIRS_LAUNCH(E, H) hipLaunchKernelGGL((k_tower_pairs<E, H>), dim3(blocks), dim3(512), 0, st, (const E*)x_hi, \
        (const E*)x_lo, ch, (E*)y_hi, (E*)y_lo, n_boards, n_dev, hd)
    if (dtype == CZ_F16) {
        if (heads) CZ_PAIRS_LAUNCH(_Float16, true); else CZ_PAIRS_LAUNCH(_Float16, false);
    } else {
        if (heads) CZ_PAIRS_LAUNCH(__bf16, true); else CZ_PAIRS_LAUNCH(__bf16, false);
    }
#undef CZ_PAIRS_LAUNCH
    if (hipGetLastError() != hipSuccess) {
        czi_set_error("cz_tower_pairs: launch failed");
        return CZ_ERR_HIP;
    }
    return CZ_OK;
}
