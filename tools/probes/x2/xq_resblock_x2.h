// xq_resblock_x2.h -- kernel 2c8x2 (round 5): the c8 / c6 residual block on TWO boards per filter fragment.
// Included by xq_conv.hip inside its anonymous namespace, after k_resblock_c8 (it uses Geom / Quad / cf8 / rb8's c6 helpers).
//
// Why.  k_resblock_c8 is not matrix-bound any more (round 4: MFMA busy 0.51): every board streams both packed filters --
// 576 KB per board and convolution -- from L2 through the CU's 64 B/clk vector-memory path, and under the chip's power cap the
// K loop's time follows the bytes it moves, not its MFMA count (tools/probes/c8_kloop_probe: the c6 loop 9.41 us per board,
// 6.49 with no loads at all; profiles/r05_kloop_probe.log).  Here a matrix wave keeps SIX pixel tiles -- two boards -- per
// filter fragment in registers (c8k::kloop<6, .., LOOK 1, RING 1>: 96 accumulators + a one-deep pixel ring fit the 256
// registers a wave has beside four copy waves), which halves the filter stream per board: 8.6 us per board in the probe.
//
// Two boards' X and Y images and an fp32 staging image do not fit 160 KB, so everything is IN PLACE: a slot (90 pixel rows x
// 256 B x 2 parts = 45 KB) holds a board's X, then its Y (epilogue 1 overwrites X once the skip operand has been folded into
// the restarted accumulators -- every element is read and written by the lane that owns it), then its output (epilogue 2, in
// the operand format of the next block: the copy waves only move bytes).  THREE slots rotate so that only one board's
// hand-over is exposed per pair:
//   pair k computes in slots (A, B); the free slot F is drained (the previous pair's second board) and refilled with the next
//   pair's first board under the K loops; after epilogue 2 the copy waves drain A and refill it with the next pair's second
//   board (held in registers) while the matrix waves wait; then (A, B, F) <- (F, A, B).
//   matrix waves:  B0 | K1(A, B) | B1 | epi1 in place, acc <- b2 + skip | B2 | K2(A, B) | B3 | epi2 in place | B4 | ...
//   copy waves:    B0 | drain F; load next A   | B1 |                        | B2 | F <- next A; load next B | B3 | | B4 | drain A; A <- next B
// Arithmetic and its order are k_resblock_c8's (the K loop's accumulation order does not depend on the tile count; the
// epilogues are the same conversions): outputs are bit-identical to cz_resblock's one-board kernel
// (tests/test_gpu_c6.py::test_two_board_block_is_bit_identical).
// Inner blocks only: the first block (fused input layer) and the last one (head convolutions from the fp32 activation)
// stay on k_resblock_c8.
namespace rb2 {
constexpr int C = 128, RB = 256, SLOT = 90, ZROW = 272, ROWS = ZROW + 16, PART = ROWS * RB, REGION = 2 * PART;
constexpr int BIAS_OFF = REGION, LDS_BYTES = BIAS_OFF + 2 * C * 4;
static_assert(3 * SLOT <= ZROW && LDS_BYTES <= 160 * 1024, "three slots + 16 zero rows + the bias vectors must fit the CU's LDS");
}  // namespace rb2

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0) -- the copy waves' HBM stores and
// prefetches, the matrix waves' filter prefetches past the end of a K loop -- and the hand-over barrier would expose a whole
// HBM round trip to the waiting matrix waves.  Nothing a barrier of this kernel orders lives in global memory.
#define CZ_X2_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); \
                              asm volatile("" ::: "memory"); } while (0)

template <bool C6>
__global__ __launch_bounds__(512, 2) void k_resblock_c8x2(
    const _Float16* __restrict__ xh, const unsigned char* __restrict__ xc, const void* __restrict__ w1p,
    const float* __restrict__ b1, const void* __restrict__ w2p, const float* __restrict__ b2, _Float16* __restrict__ yh,
    unsigned char* __restrict__ yc, int n_boards, const int32_t* __restrict__ n_dev)
{
    using namespace rb2;
    using rb8::c6_chunk;
    using rb8::pack_ints;
    typedef rb8::u32x6 u32x6;
    typedef rb8::f32x32 f32x32;
    constexpr int NT = 6, CTHR = 256, CHUNKS = 90 * 16, LITER = (CHUNKS + CTHR - 1) / CTHR;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    if (n_dev) {                                        // compact queue: the board count lives on the device
        const int nd = __builtin_amdgcn_readfirstlane(*n_dev);
        n_boards = nd < n_boards ? nd : n_boards;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int stride = gridDim.x, n_pairs = (n_boards + 1) >> 1;
    int pr = blockIdx.x;
    if (pr >= n_pairs) return;
    int sA = 0, sB = SLOT, sF = 2 * SLOT;               // first rows of the three slots (wave-uniform)

    if (wave >= 4) {                                    // ---- copy waves ----
        const int ctid = tid - 256;
        uint4 v[2][LITER];
        auto load_board = [&](int board) __attribute__((always_inline)) {
            const uint4* sh = reinterpret_cast<const uint4*>(xh + (size_t)board * 90 * C);
            const uint4* sc = reinterpret_cast<const uint4*>(xc + (size_t)board * 90 * 2 * C);
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int i = it * CTHR + ctid;
                if ((it + 1) * CTHR <= CHUNKS || i < CHUNKS) {
                    v[0][it] = sh[i];
                    v[1][it] = sc[i];
                }
            }
        };
        auto write_slot = [&](int srow) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int i = it * CTHR + ctid;
                if (!((it + 1) * CTHR <= CHUNKS || i < CHUNKS)) continue;
                const int row = i >> 4, ch = i & 15;
                unsigned char* d = lds + (srow + row) * RB + ((ch ^ (row & 15)) << 4);
                *reinterpret_cast<uint4*>(d) = v[0][it];
                *reinterpret_cast<uint4*>(d + PART) = v[1][it];
            }
        };
        // a slot's output (both parts, 16-byte chunks, whole rows) to HBM: the same chunk ownership as write_slot, so a refill
        // of the slot by the same thread needs no synchronisation beyond program order
        auto drain_slot = [&](int srow, int board) __attribute__((always_inline)) {
            uint4* dh = reinterpret_cast<uint4*>(yh + (size_t)board * 90 * C);
            uint4* dc = reinterpret_cast<uint4*>(yc + (size_t)board * 90 * 2 * C);
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int i = it * CTHR + ctid;
                if (!((it + 1) * CTHR <= CHUNKS || i < CHUNKS)) continue;
                const int row = i >> 4, ch = i & 15;
                const unsigned char* s = lds + (srow + row) * RB + ((ch ^ (row & 15)) << 4);
                const uint4 a = *reinterpret_cast<const uint4*>(s);
                const uint4 b = *reinterpret_cast<const uint4*>(s + PART);
                dh[i] = a;
                dc[i] = b;
            }
        };
        int a = 2 * pr, b = 2 * pr + 1;
        load_board(a);
        write_slot(sA);
        if (b < n_boards) {
            load_board(b);
            write_slot(sB);
        }
        for (int i = ctid; i < 16 * 16; i += CTHR) {       // the 16 all-zero rows, both parts
            *reinterpret_cast<uint4*>(lds + ZROW * RB + i * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(lds + PART + ZROW * RB + i * 16) = make_uint4(0, 0, 0, 0);
        }
        if (ctid < C) {
            reinterpret_cast<float*>(lds + BIAS_OFF)[ctid] = b1[ctid];
            reinterpret_cast<float*>(lds + BIAS_OFF)[C + ctid] = b2[ctid];
        }
        int prev_b = -1;                                   // the board whose output sits in the free slot
#ifdef CZ_RB_STAMPS
        int it_no = 0;
#endif
        for (;;) {
            const int np = pr + stride;
            const bool has_next = np < n_pairs;
            const int na = 2 * np, nb = 2 * np + 1;
            const bool has_nb = has_next && nb < n_boards;
#ifdef CZ_RB_STAMPS
            const bool stamp_on = blockIdx.x == 5 && ctid == 0 && it_no++ == 20;
#endif
            RB_STAMP(16);
            CZ_X2_BARRIER();                               // B0: slots A, B hold the pair's X
            RB_STAMP(17);
            if (prev_b >= 0) drain_slot(sF, prev_b);
            if (has_next) load_board(na);
            RB_STAMP(18);
            CZ_X2_BARRIER();                               // B1
            CZ_X2_BARRIER();                               // B2
            RB_STAMP(19);
            if (has_next) write_slot(sF);
            if (has_nb) load_board(nb);
            RB_STAMP(20);
            CZ_X2_BARRIER();                               // B3
            CZ_X2_BARRIER();                               // B4: both outputs are in place
            RB_STAMP(21);
            drain_slot(sA, a);
            RB_STAMP(22);
            if (has_nb) write_slot(sA);
            RB_STAMP(23);
            prev_b = b < n_boards ? b : -1;
            if (!has_next) break;
            const int t = sA;                              // (A, B, F) <- (F, A, B)
            sA = sF;
            sF = sB;
            sB = t;
            a = na;
            b = nb;
            pr = np;
        }
        if (prev_b >= 0) drain_slot(sB, prev_b);
        return;
    }

    // ---- matrix waves ----
    const int kb = lane >> 5, ln = lane & 31;
    const c8k::Filter flt1 = c8k::make_filter(w1p, wave, lane), flt2 = c8k::make_filter(w2p, wave, lane);
    const float* bias1 = reinterpret_cast<const float*>(lds + BIAS_OFF);
    const float* bias2 = bias1 + C;
    const int k_x = C6 ? __builtin_amdgcn_readfirstlane(pack_ints(w1p)[2]) : 0;     // exponents of the images read / written
    const int k_y = C6 ? __builtin_amdgcn_readfirstlane(pack_ints(w2p)[2]) : 0;
    const int k_o = C6 ? __builtin_amdgcn_readfirstlane(pack_ints(w2p)[3]) : 0;
    f32x16 acc[NT];

    // relu(acc) of one board (tiles tb .. tb + 2) -> the slot at row srow, in the operand format with image exponent k_w;
    // SKIP: first fold this lane's own elements of the image the slot holds (exponent k_s) and bias2 into the accumulators'
    // restart values -- read before anything of the same (tile, channel block) is overwritten.
    auto epilogue = [&](int tb, int srow, int k_w, int k_s, bool skip) __attribute__((always_inline)) {
        unsigned char* img = lds + srow * RB;
        if (C6) {
            const float s_hi = __builtin_ldexpf(1.0f, k_w), s_lo = __builtin_ldexpf(1.0f, k_w - cf8::X_LO_SHIFT);
            const float s_skip = __builtin_ldexpf(1.0f, k_s - cf8::X_LO_SHIFT);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {               // tiles (0, 1) trade lane halves, tile 2 trades with itself
                const int pa = tb + (pp == 0 ? 0 : 2), pb = tb + (pp == 0 ? 1 : 2);
                f32x16 sk[2];
                if (skip) {
#pragma unroll
                    for (int h = 0; h < (pp == 0 ? 2 : 1); ++h) {
                        const int q = (pp == 0 ? h : 2) * 32 + ln;
                        const int row = q < 90 ? q : 89;
                        const uint4 hd4 = *reinterpret_cast<const uint4*>(img + PART + rb8::c6_lds_off(row, c6_chunk(0, wave)));
                        const uint2 tl2 = *reinterpret_cast<const uint2*>(img + PART + rb8::c6_lds_off(row, c6_chunk(0, wave) + 1));
                        const uint32_t wv[7] = {hd4.x, hd4.y, hd4.z, hd4.w, tl2.x, tl2.y, 0u};
                        const uint32_t sh6 = (uint32_t)kb * 6u;
                        u32x6 pc;
#pragma unroll
                        for (int w = 0; w < 6; ++w) pc[w] = __builtin_amdgcn_alignbit(wv[w + 1], wv[w], sh6);
                        const f32x32 xl = __builtin_amdgcn_cvt_scalef32_pk32_f32_bf6(pc, s_skip);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int ch = wave * 32 + g * 8 + kb * 4;
                            const int off = row * RB + (((ch >> 3) ^ (row & 15)) << 4) + (ch & 7) * 2;
                            const float4 bv4 = *reinterpret_cast<const float4*>(bias2 + ch);
                            float vv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
                            const Quad<_Float16> xq = *reinterpret_cast<const Quad<_Float16>*>(img + off);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r = g * 4 + i;
                                vv[i] += (float)xq.e[i] + xl[2 * r];
                                sk[h][r] = vv[i];
                            }
                        }
                    }
                }
                f32x16 lo[2];
#pragma unroll
                for (int h = 0; h < (pp == 0 ? 2 : 1); ++h) {
                    const int p = pp == 0 ? pa + h : pa;
                    const int q = (pp == 0 ? h : 2) * 32 + ln;
                    const int row = q < 90 ? q : 89;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = wave * 32 + g * 8 + kb * 4;
                        const int off = row * RB + (((ch >> 3) ^ (row & 15)) << 4) + (ch & 7) * 2;
                        Quad<_Float16> hq;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float r = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                            hq.e[i] = (_Float16)r;
                            acc[p][g * 4 + i] = r;
                            lo[h][g * 4 + i] = r - (float)hq.e[i];
                        }
                        if (q < 90) *reinterpret_cast<Quad<_Float16>*>(img + off) = hq;
                    }
                }
                {
                    const int hb = pp == 0 ? 1 : 0;        // (tile 2: both operands of the swap are the same tile)
                    f32x16 av, bv, al, bl;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[pa][r]), __float_as_uint(acc[pb][r]), false, false);
                        const auto sl = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[0][r]), __float_as_uint(lo[hb][r]), false, false);
                        av[r] = __uint_as_float(sv[0]); bv[r] = __uint_as_float(sv[1]);
                        al[r] = __uint_as_float(sl[0]); bl[r] = __uint_as_float(sl[1]);
                    }
                    const u32x6 pl = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(al, bl, s_lo);
                    const u32x6 pv = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(av, bv, s_hi);
                    const int q = pp == 0 ? kb * 32 + ln : 64 + ln;
                    if (pp == 0 || (kb == 0 && q < 90)) {
                        unsigned char* d0 = img + PART + rb8::c6_lds_off(q, c6_chunk(0, wave));
                        unsigned char* t0 = img + PART + rb8::c6_lds_off(q, c6_chunk(0, wave) + 1);
                        unsigned char* d1 = img + PART + rb8::c6_lds_off(q, c6_chunk(1, wave));
                        unsigned char* t1 = img + PART + rb8::c6_lds_off(q, c6_chunk(1, wave) + 1);
                        *reinterpret_cast<uint4*>(d0) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                        *reinterpret_cast<uint2*>(t0) = make_uint2(pl[4], pl[5]);
                        *reinterpret_cast<uint4*>(d1) = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                        *reinterpret_cast<uint2*>(t1) = make_uint2(pv[4], pv[5]);
                    }
                }
                if (skip) {
                    acc[pa] = sk[0];
                    if (pp == 0) acc[pb] = sk[1];
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int p = tb + t;
                const int q = t * 32 + ln;
                const int row = q < 90 ? q : 89;           // (padding lanes compute on row 89 and store nothing)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = wave * 32 + g * 8 + kb * 4;
                    const int off = row * RB + (((ch >> 3) ^ (row & 15)) << 4) + (ch & 7) * 2;
                    const int off_lo = PART + row * RB + (ch & 15) + (((ch >> 4) ^ (row & 15)) << 4);
                    const int off_hi = PART + row * RB + (ch & 15) + (((8 + (ch >> 4)) ^ (row & 15)) << 4);
                    float vv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (skip) {
                        const float4 bv = *reinterpret_cast<const float4*>(bias2 + ch);
                        vv[0] = bv.x; vv[1] = bv.y; vv[2] = bv.z; vv[3] = bv.w;
                        cf8::add_pair4(vv, *reinterpret_cast<const Quad<_Float16>*>(img + off), *reinterpret_cast<const uint32_t*>(img + off_lo));
                    }
                    float r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = acc[p][g * 4 + i] > 0.0f ? acc[p][g * 4 + i] : 0.0f;
                    const cf8::Split4 o = cf8::split4(r);
                    if (q < 90) {
                        *reinterpret_cast<Quad<_Float16>*>(img + off) = o.hi;
                        *reinterpret_cast<uint32_t*>(img + off_lo) = o.l8;
                        *reinterpret_cast<uint32_t*>(img + off_hi) = o.h8;
                    }
                    if (skip) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[p][g * 4 + i] = vv[i];
                    }
                }
            }
        }
    };

#ifdef CZ_RB_STAMPS
    int it_no = 0;
#endif
    for (;;) {
#ifdef CZ_RB_STAMPS
        const bool stamp_on = blockIdx.x == 5 && tid == 0 && it_no++ == 20;
#endif
        const bool has_next = pr + stride < n_pairs;
        RB_STAMP(0);
        CZ_X2_BARRIER();                                   // B0
        RB_STAMP(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(bias1 + wave * 32 + g * 8 + kb * 4);
#pragma unroll
            for (int p = 0; p < NT; ++p) {
                acc[p][g * 4 + 0] = bv.x; acc[p][g * 4 + 1] = bv.y; acc[p][g * 4 + 2] = bv.z; acc[p][g * 4 + 3] = bv.w;
            }
        }
        __builtin_amdgcn_s_setprio(3);
        c8k::kloop<NT, c8k::NoShadow, 0, false, 128, C6 ? 1 : 0, 1, 1, true>(
            lds, c8k::Image{sA, ZROW, PART, sB}, flt1, lane, acc, 127 + k_x - cf8::X_LO_SHIFT, 127 + k_x);
        __builtin_amdgcn_s_setprio(0);
        RB_STAMP(2);
        CZ_X2_BARRIER();                                   // B1: every wave is done reading X
        RB_STAMP(3);
        epilogue(0, sA, k_y, k_x, true);
        epilogue(3, sB, k_y, k_x, true);
        RB_STAMP(4);
        CZ_X2_BARRIER();                                   // B2: Y complete
        RB_STAMP(5);
        __builtin_amdgcn_s_setprio(3);
        c8k::kloop<NT, c8k::NoShadow, 0, false, 128, C6 ? 1 : 0, 1, 1, true>(
            lds, c8k::Image{sA, ZROW, PART, sB}, flt2, lane, acc, 127 + k_y - cf8::X_LO_SHIFT, 127 + k_y);
        __builtin_amdgcn_s_setprio(0);
        RB_STAMP(6);
        CZ_X2_BARRIER();                                   // B3: every wave is done reading Y
        RB_STAMP(7);
        epilogue(0, sA, k_o, 0, false);
        epilogue(3, sB, k_o, 0, false);
        RB_STAMP(8);
        CZ_X2_BARRIER();                                   // B4: the outputs are in place
        RB_STAMP(9);
        if (!has_next) break;
        const int t = sA;
        sA = sF;
        sF = sB;
        sB = t;
        pr += stride;
    }
}
