// xq_c8_kloop.h -- the K loop of the c8 tower arithmetic (round 4) and of c6, its sibling with bf6 correction operands
// (FMT = 1, at the kloop template below), shared by csrc/xq_conv.hip (k_conv3x3_c8, the c8 / c6 residual-block kernels)
// and tools/probes/c8_kloop_probe.hip (which times exactly this instruction stream).
//
// One call = 9 taps x 128 input channels for the 32 output channels of this wave and NT pixel tiles of 32 of an LDS
// image; per 64-channel block  two fp16 K-steps -> the block's e4m3(w) x_lo8 MFMAs (K = 64) -> two fp16 K-steps -> its
// w_lo8 e4m3(x) MFMAs: the accumulation order of round 3's loop, so results are bit-identical to it.
//
// What changed against round 3's conv_kloop_c8 (3.34 ms per block launch, MFMA busy 0.55: the loop ran at 59 % of its
// MFMA floor, profiles/r03_c8_kloop_probe.log) -- read off its ISA:
//   * every second fp16 K-step waited with lgkmcnt(2): its pixel fragments had been requested ONE K-step (96 cycles)
//     earlier, less than a loaded LDS round trip.  Fragments are now requested TWO K-steps ahead (ring of three).
//   * an fp16 MFMA slot is 32 cycles = 8 issue slots of which ~5 are usable (MI355X_MICROARCH.md); round 3 put the next
//     tap's row arithmetic (~10 VALU), the 64-bit address arithmetic of the filter loads (5 VALU + s_nop per pair) and
//     the loads themselves into fp16 slots -- 9 to 15 instructions in one slot of three.  Now an fp16 slot carries the
//     MFMA, one ds_read_b128, one v_xor (and, twice per half block, one buffer_load whose address is an SGPR); everything
//     else sits in the 64-cycle fp8 slots, which have twice the room.
//   * filter fragments come through buffer_load_dwordx4 with a wave-uniform SGPR offset: no per-lane 64-bit adds.
//   * the c8 filter pieces are single-buffered (16 registers instead of 32): a kind's pieces are re-requested right after
//     the fp8 MFMAs that used them have issued, one block (768 cycles) before their next use.
//   * the c8 PIXEL pieces of one kind only are live at a time (8 NT registers instead of 16 NT): the 2 NT pieces a half
//     block's correction MFMAs need are read in the 2 NT fp16 slots right before them, tile 0 first (>= 128 cycles of
//     lead each); the fp8 slots are left with the MFMA, the filter loads, the next tap's row arithmetic -- and room for a
//     caller's shadow work (k_resblock_c8's deferred second epilogue).  (Reading them in the fp8 slots of the other kind,
//     one half block ahead, measured the same on the plain block and 12 % worse with shadow work in those slots.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// CZ_C6_TAIL_SWZ (round 5 experiment, default OFF): a c6 piece's 8-byte tail in the UPPER half of its 16-byte chunk when bit 4 of
// the pixel row is set.  ds_read_b64 serves 32 lanes per pass; pixel rows r and r + 16 carry the same chunk swizzle, so with
// every tail in the lower half lanes ln and ln + 16 of a pass hit the same two banks (a two-way conflict on every tail read:
// LDS bank-conflict fraction 0.14 against c8's 0.068, profiles/r04_pmc_nn.json); with the half chosen by row bit 4 the 32 lanes
// cover all 64 banks.  Bit-identical results (tests/test_gpu_c6.py on both builds) -- and 0.7 % SLOWER in the bench, three
// alternating pairs on one box (profiles/r05_ab_c6_tail_swizzle.log: 2.960 -> 2.983 ms per block launch): the two VALU
// instructions per tail read cost more issue slots of a 32-cycle bf6 slot than the conflicts cost LDS passes that sit in the
// MFMAs' shadow anyway.  Kept as a build switch; one convention for LDS and HBM (images are copied in 16-byte chunks).
#ifndef CZ_C6_TAIL_SWZ
#define CZ_C6_TAIL_SWZ 0
#endif

namespace c8k {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int W_PAD_STEPS = 3;
constexpr int X_LO_SHIFT = 11;
// geometry of a tower of CH filters (128: 256-byte pixel rows, 16 chunks swizzled by the row's low four bits; 192: 384-byte
// rows, 24 chunks, swizzled inside aligned groups of eight so that the XOR stays inside the row)
template <int CH> struct Geo {
    static constexpr int C = CH, RB = 2 * CH, CPR = RB / 16, KK = CH / 16, NB = CH / 64, CT = CH / 32;
    static constexpr bool POW2 = (RB & (RB - 1)) == 0;
    static constexpr int SWZ = POW2 ? 15 : 7;
    static constexpr int MAIN_U4 = (9 * KK + W_PAD_STEPS) * CT * 64;      // uint4 of the fp16 fragments of a packed filter
    static constexpr int C8_U4 = (9 * NB + 1) * 2 * CT * 2 * 64;          // ... of its e4m3 fragments (one zero block appended)
    static_assert(CH % 64 == 0 && KK % 4 == 0 && (3 * KK) % 3 == 0 && (POW2 || CPR % 8 == 0), "128 / 192 filters");
};
constexpr int C = 128, MAIN_U4 = Geo<128>::MAIN_U4, C8_U4 = Geo<128>::C8_U4;       // (the 128-filter tower's, for its callers)

// where an image sits in LDS: byte offsets of its first pixel row and of the 16 all-zero rows inside a part, and the
// distance between the fp16 part and the c8 part
struct Image {
    int row_base;       // first row of the image (rows are RB bytes)
    int zrow;           // first of the 16 zero rows (a multiple of 16)
    int part_bytes;     // c8 part = fp16 part + part_bytes
    int row_base2 = 0;  // SLOTS: first row of the SECOND board's image (pixel tiles 3 .. 5); the swizzle key is then the
                        // board-relative pixel, so an image may start at any row
};

// A packed filter as the loop reads it: a buffer resource over the whole packed tensor and this lane's byte offset
// (wave * 1024 + lane * 16 for the fp16 fragments; the c8 pieces add their own base).
struct Filter {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_main;      // byte offset of this lane inside a K-step of fp16 fragments
    int lane_c8;        // byte offset of this lane inside a (block, kind) group of c8 pieces, counted from the c8 base
    int lane_c6t;       // c6 pieces (24 bytes): the 8-byte tails sit densely behind the 16-byte heads of the group
    int scale_w_hi, scale_w_lo;      // E8M0 scale bytes of the two correction MFMAs (127 - shift), per lane: this row's
};

template <int CH = 128>
__device__ __forceinline__ Filter make_filter(const void* packed, int wave, int lane)
{
    constexpr int MAIN_U4 = Geo<CH>::MAIN_U4, C8_U4 = Geo<CH>::C8_U4;
    Filter f;
    f.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(packed), 0, (MAIN_U4 + C8_U4 + 1) * 16, 0x00020000);
    f.lane_main = wave * 1024 + lane * 16;
    f.lane_c8 = MAIN_U4 * 16 + wave * 2048 + lane * 16;
    f.lane_c6t = MAIN_U4 * 16 + wave * 2048 + 1024 + lane * 8;
    // the correction operands' shifts, one pair per OUTPUT CHANNEL (the matrix instruction reads the A operand's E8M0 scale
    // per lane = per row): 2 x CH signed bytes behind the 16-byte block of ints, this lane's row = wave * 32 + (lane & 31)
    const signed char* rs = reinterpret_cast<const signed char*>(reinterpret_cast<const uint4*>(packed) + MAIN_U4 + C8_U4 + 1);
    const int row = wave * 32 + (lane & 31);
    f.scale_w_hi = 127 - (int)rs[row];
    f.scale_w_lo = 127 - (int)rs[CH + row];
    return f;
}

// Hooks: work another part of a kernel wants done in the shadow of the MFMAs (a residual block's second epilogue).
// fp8(j, slot) is called once in every fp8 slot: j = 0 .. 2 is the (rolled) loop iteration, slot = 0 .. 12 NT - 1 the slot's
// position inside it -- a compile-time constant after unrolling, so that register indices derived from it are static.
struct NoShadow {
    __device__ __forceinline__ void fp8(int, int) {}
};

// PROBE (tools/probes only; 0 in the product): bit 0 = no filter loads inside the loop (the prologue's fragments are reused),
// bit 1 = no LDS reads inside the loop -- timing experiments with wrong results.
// ZERO_INIT = false: the caller has put the accumulators' start values into acc (bias, bias + skip connection): the
// products are added on top, so the epilogue that follows has no additions left to do.
// FMT = 1: the c6 arithmetic -- correction operands in bf6 (e3m2), 24-byte pieces (a 16-byte head where the e4m3 piece's
// first half sits, an 8-byte tail at the start of its second half), correction MFMAs of 32 cycles instead of 64: the loop's
// matrix work is 3/4 of c8's, and the room the long fp8 slots offered is gone -- the next tap's row arithmetic is cut
// into three stages of 3-4 instructions, one per correction slot of the tap's first three half blocks.
// LOOK / RING: pixel fragments are requested LOOK K-steps ahead into a ring of RING slots.  (2, 3) for three pixel tiles (a
// K-step = 96 cycles); (1, 1) for six -- two boards per filter fragment (round 5 experiment, tools/probes/x2/): a K-step is 192 cycles, the
// request for tile i of the next K-step is issued right after this K-step's MFMA on tile i has read the same registers, and the
// ring shrinks from 12 NT to 4 NT registers (six tiles have to live in 256 registers beside the 96 accumulators).
// SLOTS: the two boards' images start at img.row_base and img.row_base2 (any rows) instead of 90 rows apart.
template <int NT, typename Shadow = NoShadow, int PROBE = 0, bool ZERO_INIT = true, int CH = 128, int FMT = 0, int LOOK = 2,
          int RING = 3, bool SLOTS = false>
__device__ __forceinline__ void kloop(const unsigned char* lds, const Image img, const Filter& flt, int lane, f32x16* acc,
                                      int scale_x_lo, int scale_x, Shadow&& shadow = NoShadow())
{
    typedef Geo<CH> G;
    constexpr int RB = G::RB, CPR = G::CPR, SWZ = G::SWZ, KK = G::KK, NB = G::NB, CT = G::CT;
    const int kb = lane >> 5, ln = lane & 31;
    int qy[3], qx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = t * 32 + ln;
        qy[t] = q < 90 ? q / 9 : 100;                  // 100: never on the board, whatever the tap
        qx[t] = q - (q / 9) * 9;
    }
    // byte offset (within a part) of this lane's fp16 fragment for K-step 0 of tap (dy, dx), pixel tile p
    auto tap_row = [&](int dy, int dx, int p) {
        const int t = p % 3;
        const bool ok = (unsigned)(qy[t] + dy) < 10u && (unsigned)(qx[t] + dx) < 9u;
        const int nominal = (SLOTS ? 0 : (p / 3) * 90) + t * 32 + ln + dy * 9 + dx;
        const int row = ok ? (SLOTS && p >= 3 ? img.row_base2 : img.row_base) + nominal : img.zrow + (nominal & 15);
        return row * RB + (((kb ^ nominal) & SWZ) << 4);
    };
    const int lane_c = (kb * 3) << 4;
    // fp16 fragment of K-step kk: chunk 2 kk + kb of the row, swizzled; pre_p already holds the lane's (kb ^ row) bits
    auto load_px = [&](int pre_p, int kk) {
        const int off = G::POW2 ? pre_p ^ (kk << 5) : (pre_p ^ ((kk & 3) << 5)) + ((kk >> 2) << 7);
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(lds + off));
    };
    // c8 piece h of (kind q, block b): chunk c0 + 2 kb of the row, c0 = q * CPR/2 + 4 b + h (bit 1 of c0 is never set)
    auto load_c8 = [&](i32x8& d, int pre_p, int q, int b, int h) {
        const int c0 = q * (CPR / 2) + 4 * b + h;
        const int off = G::POW2 ? pre_p ^ lane_c ^ (c0 << 4) : (pre_p ^ lane_c ^ ((c0 & 7) << 4)) + ((c0 >> 3) << 7);
        if (FMT == 0 || h == 0) {
            const uint4 t = *reinterpret_cast<const uint4*>(lds + img.part_bytes + off);
            d[4 * h + 0] = t.x; d[4 * h + 1] = t.y; d[4 * h + 2] = t.z; d[4 * h + 3] = t.w;
        } else {
            // (rows of an image start at a multiple of 32 for every c6 caller, so bit 4 of the pixel row is address bit 12 of
            //  pre_p; the zero rows hold zeros in both halves)
            const int half = CZ_C6_TAIL_SWZ ? (pre_p >> 9) & 8 : 0;
            const uint2 t = *reinterpret_cast<const uint2*>(lds + img.part_bytes + off + half);
            d[4] = t.x; d[5] = t.y;
        }
    };
    constexpr int STEP_B = CT * 1024, BLK_B = 2 * CT * 2048;     // bytes of a K-step of fp16 fragments / of a block's c8 pieces
    auto load_w = [&](int step_soff) {                 // fp16 fragment of K-step `step` (soffset = step * STEP_B, wave-uniform)
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(flt.rsrc, flt.lane_main, step_soff, 0));
    };
    auto load_wc = [&](i32x8& d, int blk_soff, int q, int h) {      // blk_soff = blk * BLK_B (one block = 2 kinds x CT waves x 2 KB)
        if (FMT == 0 || h == 0) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(flt.rsrc, flt.lane_c8 + h * 1024, blk_soff + q * (BLK_B / 2), 0);
            d[4 * h + 0] = (int)t.x; d[4 * h + 1] = (int)t.y; d[4 * h + 2] = (int)t.z; d[4 * h + 3] = (int)t.w;
        } else {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(flt.rsrc, flt.lane_c6t, blk_soff + q * (BLK_B / 2), 0);
            d[4] = (int)t.x; d[5] = (int)t.y;
        }
    };
    // the next tap's rows in three stages (FMT = 1): A nominal pixel + on-board test, B row, C byte offset
    int st_nom[NT], st_row[NT];
    bool st_ok[NT];
    auto tap_stage = [&](int stage, int dy, int dx, int p, int* out) {
        const int t = p % 3;
        if (stage == 0) {
            st_ok[p] = (unsigned)(qy[t] + dy) < 10u && (unsigned)(qx[t] + dx) < 9u;
            st_nom[p] = (SLOTS ? 0 : (p / 3) * 90) + t * 32 + ln + dy * 9 + dx;
        } else if (stage == 1) {
            st_row[p] = st_ok[p] ? (SLOTS && p >= 3 ? img.row_base2 : img.row_base) + st_nom[p] : img.zrow + (st_nom[p] & 15);
        } else {
            out[p] = st_row[p] * RB + (((kb ^ st_nom[p]) & SWZ) << 4);
        }
    };

    f16x8 wf[4];                                        // fp16 filter fragments, slot = K-step % 4
    static_assert(24 % RING == 0 && LOOK >= 1 && LOOK <= RING + (RING == 1) && LOOK <= 2, "ring indices are static per iteration");
    f16x8 px[RING][NT];                                 // pixel fragments, slot = K-step % RING (24 K-steps per loop iteration)
    i32x8 cx[NT];                                       // c8 pixel pieces of the kind whose MFMAs come next
    i32x8 wcr[2];                                       // c8 filter pieces by kind
    int pre[NT], pre_n[NT];
    if (FMT == 1) {                                     // (registers 6, 7 of a bf6 operand are not read)
#pragma unroll
        for (int p = 0; p < NT; ++p) { cx[p][6] = 0; cx[p][7] = 0; }
        wcr[0][6] = wcr[0][7] = wcr[1][6] = wcr[1][7] = 0;
    }
    if (ZERO_INIT) {
#pragma unroll
        for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;
    }
#pragma unroll
    for (int p = 0; p < NT; ++p) pre[p] = tap_row(-1, -1, p);
#pragma unroll
    for (int s = 0; s < 4; ++s) wf[s] = load_w(s * STEP_B);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        load_wc(wcr[q], 0, q, 0);
        load_wc(wcr[q], 0, q, 1);
    }
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        px[0][p] = load_px(pre[p], 0);
        if (LOOK == 2) px[1 % RING][p] = load_px(pre[p], 1);
    }

#pragma unroll 1
    for (int j = 0; j < 3; ++j) {                       // taps 3 j .. 3 j + 2 (dy = j - 1)
        const int soff_j = j * (3 * KK * STEP_B), boff_j = j * (3 * NB * BLK_B);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int tap3 = tt;                         // tap inside the iteration
            // the NEXT tap (the last one: itself; its prefetches read valid rows and are never used)
            const int ndy = tt < 2 ? j - 1 : (j < 2 ? j : 1), ndx = tt < 2 ? tt : (j < 2 ? -1 : 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int kk = b * 4 + half * 2 + k2;                 // K-step inside the tap
                        const int g = tap3 * KK + kk;                         // K-step inside the iteration (ring indices)
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk & 3], px[g % RING][i], acc[i], 0, 0, 0);
                            const int kn = kk + LOOK;                          // fragments of the K-step after next (LOOK = 2)
                            if (!(PROBE & 2)) {
                                px[(g + LOOK) % RING][i] = kn < KK ? load_px(pre[i], kn) : load_px(pre_n[i], kn - KK);
                                // piece (k2 NT + i) of the 2 NT c8 pixel pieces of this half block's correction MFMAs
                                const int s6 = k2 * NT + i;
                                load_c8(cx[s6 >> 1], pre[s6 >> 1], half, b, s6 & 1);
                            }
                            // the c8 filter pieces of the kind whose MFMAs have just issued, for its next block
                            if (i == 0 && !(PROBE & 1)) {
                                const int q = 1 - half;                        // kind used in the PREVIOUS fp8 group
                                const int blk_next = tap3 * NB + b + (half == 0 ? 0 : 1);     // its next block (half 0: kind 1 of b - 1 -> b)
                                load_wc(wcr[q], boff_j + blk_next * BLK_B, q, k2);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // ---- correction term `half` of this block (K = 64)
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wcr[half], cx[i], acc[i], FMT ? 3 : 0, FMT ? 3 : 0, 0,
                                                                                  half ? flt.scale_w_lo : flt.scale_w_hi, 0,
                                                                                  half ? scale_x : scale_x_lo);
                        // the fp16 filter fragments of the two K-steps just retired, one block ahead (their ring slots are free)
                        if (i < 2 && !(PROBE & 1)) {
                            const int kk = b * 4 + half * 2 + i;
                            wf[kk & 3] = load_w(soff_j + (tap3 * KK + kk + 4) * STEP_B);
                        }
                        // the next tap's rows: one per fp8 slot of the tap's first block (needed from K-step 6 on)
                        if (FMT == 0) {
                            if (b == 0 && half == 0) pre_n[i] = tap_row(ndy, ndx, i);
                        } else if (NT == 6) {
                            // six tiles: a tile's three stages in three consecutive slots of one group, two tiles per group -- the
                            // stage registers are live for three slots of one tile instead of two groups of all six
                            if (b * 2 + half < 3) tap_stage(i % 3, ndy, ndx, (b * 2 + half) * 2 + i / 3, pre_n);
                        } else if (b * 2 + half < 3) {
                            tap_stage(b * 2 + half, ndy, ndx, i, pre_n);
                        }
                        shadow.fp8(j, ((tt * NB + b) * 2 + half) * NT + i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) pre[p] = pre_n[p];
        }
    }
}


// ---- the same K loop for CTW channel tiles per wave (round 6; k_resblock_ip4_c8: 192 filters, four matrix waves of three
// channel tiles each, one wave per SIMD = 512 registers).  Per accumulator tile the products arrive in exactly kloop's order
// (per 64-channel block: two fp16 K-steps, correction term 0, two fp16 K-steps, correction term 1), so results are bit-identical
// to it; what changes is the traffic: a pixel fragment read from LDS feeds CTW MFMAs, and six channel tiles spread evenly over
// four SIMDs (with one tile per wave two SIMDs carry two matrix waves and two carry one).  flt[c]: the wave's c-th channel tile
// (make_filter with the tile index in place of the wave index); acc[c * NT + i]: channel tile c, pixel tile i.
template <int CTW, int NT, int CH, int FMT>
__device__ __forceinline__ void kloop_ctw(const unsigned char* lds, const Image img, const Filter* flt, int lane, f32x16* acc,
                                          int scale_x_lo, int scale_x)
{
    typedef Geo<CH> G;
    constexpr int RB = G::RB, CPR = G::CPR, SWZ = G::SWZ, KK = G::KK, NB = G::NB, CT = G::CT;
    const int kb = lane >> 5, ln = lane & 31;
    int qy[3], qx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int q = t * 32 + ln;
        qy[t] = q < 90 ? q / 9 : 100;
        qx[t] = q - (q / 9) * 9;
    }
    auto tap_row = [&](int dy, int dx, int p) {
        const int t = p % 3;
        const bool ok = (unsigned)(qy[t] + dy) < 10u && (unsigned)(qx[t] + dx) < 9u;
        const int nominal = (p / 3) * 90 + t * 32 + ln + dy * 9 + dx;
        const int row = ok ? img.row_base + nominal : img.zrow + (nominal & 15);
        return row * RB + (((kb ^ nominal) & SWZ) << 4);
    };
    const int lane_c = (kb * 3) << 4;
    auto load_px = [&](int pre_p, int kk) {
        const int off = G::POW2 ? pre_p ^ (kk << 5) : (pre_p ^ ((kk & 3) << 5)) + ((kk >> 2) << 7);
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(lds + off));
    };
    auto load_c8 = [&](i32x8& d, int pre_p, int q, int b, int h) {
        const int c0 = q * (CPR / 2) + 4 * b + h;
        const int off = G::POW2 ? pre_p ^ lane_c ^ (c0 << 4) : (pre_p ^ lane_c ^ ((c0 & 7) << 4)) + ((c0 >> 3) << 7);
        if (FMT == 0 || h == 0) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(lds + img.part_bytes + off);
            d[4 * h + 0] = t.x; d[4 * h + 1] = t.y; d[4 * h + 2] = t.z; d[4 * h + 3] = t.w;
        } else {
            const u32x2 t = *reinterpret_cast<const u32x2*>(lds + img.part_bytes + off);
            d[4] = t.x; d[5] = t.y;
        }
    };
    constexpr int STEP_B = CT * 1024, BLK_B = 2 * CT * 2048;
    auto load_w = [&](int c, int step_soff) {
        return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(flt[c].rsrc, flt[c].lane_main, step_soff, 0));
    };
    auto load_wc = [&](i32x8& d, int c, int blk_soff, int q, int h) {
        if (FMT == 0 || h == 0) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(flt[c].rsrc, flt[c].lane_c8 + h * 1024, blk_soff + q * (BLK_B / 2), 0);
            d[4 * h + 0] = (int)t.x; d[4 * h + 1] = (int)t.y; d[4 * h + 2] = (int)t.z; d[4 * h + 3] = (int)t.w;
        } else {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(flt[c].rsrc, flt[c].lane_c6t, blk_soff + q * (BLK_B / 2), 0);
            d[4] = (int)t.x; d[5] = (int)t.y;
        }
    };
    f16x8 wf[4][CTW];
    f16x8 px[3][NT];
    i32x8 cx[NT];
    i32x8 wcr[2][CTW];
    int pre[NT], pre_n[NT];
    if (FMT == 1) {
#pragma unroll
        for (int p = 0; p < NT; ++p) { cx[p][6] = 0; cx[p][7] = 0; }
#pragma unroll
        for (int c = 0; c < CTW; ++c) { wcr[0][c][6] = wcr[0][c][7] = wcr[1][c][6] = wcr[1][c][7] = 0; }
    }
#pragma unroll
    for (int p = 0; p < NT; ++p) pre[p] = tap_row(-1, -1, p);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < CTW; ++c) wf[s][c] = load_w(c, s * STEP_B);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            load_wc(wcr[q][c], c, 0, q, 0);
            load_wc(wcr[q][c], c, 0, q, 1);
        }
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        px[0][p] = load_px(pre[p], 0);
        px[1][p] = load_px(pre[p], 1);
    }
    static_assert((3 * KK) % 3 == 0, "ring indices are static per iteration");
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        const int soff_j = j * (3 * KK * STEP_B), boff_j = j * (3 * NB * BLK_B);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int ndy = tt < 2 ? j - 1 : (j < 2 ? j : 1), ndx = tt < 2 ? tt : (j < 2 ? -1 : 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int kk = b * 4 + half * 2 + k2;
                        const int g = tt * KK + kk;
#pragma unroll
                        for (int i = 0; i < NT; ++i) {
#pragma unroll
                            for (int c = 0; c < CTW; ++c)
                                acc[c * NT + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk & 3][c], px[g % 3][i], acc[c * NT + i], 0, 0, 0);
                            const int kn = kk + 2;
                            px[(g + 2) % 3][i] = kn < KK ? load_px(pre[i], kn) : load_px(pre_n[i], kn - KK);
                            const int s6 = k2 * NT + i;                    // piece s6 of the 2 NT pixel pieces of this half block
                            load_c8(cx[s6 >> 1], pre[s6 >> 1], half, b, s6 & 1);
                            if (i < CTW) {                                 // the filter pieces of the kind used one group ago, next block
                                const int q = 1 - half;
                                const int blk_next = tt * NB + b + (half == 0 ? 0 : 1);
                                load_wc(wcr[q][i], i, boff_j + blk_next * BLK_B, q, k2);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
#pragma unroll
                        for (int c = 0; c < CTW; ++c)
                            acc[c * NT + i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                wcr[half][c], cx[i], acc[c * NT + i], FMT ? 3 : 0, FMT ? 3 : 0, 0,
                                half ? flt[c].scale_w_lo : flt[c].scale_w_hi, 0, half ? scale_x : scale_x_lo);
                        if (i < 2) {                                       // the fp16 fragments of the two K-steps just retired, one block ahead
                            const int kk = b * 4 + half * 2 + i;
#pragma unroll
                            for (int c = 0; c < CTW; ++c) wf[kk & 3][c] = load_w(c, soff_j + (tt * KK + kk + 4) * STEP_B);
                        }
                        if (b == 0 && half == 0) pre_n[i] = tap_row(ndy, ndx, i);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < NT; ++p) pre[p] = pre_n[p];
        }
    }
}

}  // namespace c8k
