// xq_lane.h -- per-lane (per-square) pieces of the Xiangqi rules.
//
// These are the scalar building blocks each lane of a wavefront executes; the wave-level
// composition (ordered compaction by prefix sum, ballots) lives in xq_rules.h.  They are
// XQ_HD so the CPU-side unit test (tests/lane_harness.cpp) can run them square by square
// and compare with the oracle before anything touches a GPU.
//
// Semantics follow the reference's get_legal_moves (static_env.py:256-348) including its
// per-piece emission order (mov_dir, light_env/common.py:66-76).
#pragma once
#include "xq_tables.h"

#if defined(__HIPCC__)
#define XQ_HD __host__ __device__ __forceinline__
#define XQ_D __device__ __forceinline__
#else
#define XQ_HD inline
#endif

namespace xq {

#if defined(__HIPCC__)
static __device__ const Tables d_tab = make_tables();
#endif
static constexpr Tables h_tab = make_tables();
static_assert(h_tab.lab_ft[NLABELS] == NLABELS, "label table must have exactly 2086 entries");

XQ_HD uint16_t label_of(int from, int to)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return d_tab.label_of[from * NSQ + to];
#else
    return h_tab.label_of[from * NSQ + to];
#endif
}
XQ_HD uint16_t label_ft(int label)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return d_tab.lab_ft[label];
#else
    return h_tab.lab_ft[label];
#endif
}

// Sink for generated moves: an ordered list of (label, from<<8|to) starting at `off`.
// EMIT=false only counts (first pass of the two-pass ordered compaction).
template <bool EMIT>
struct MoveSink {
    uint16_t* lab;
    uint16_t* ft;
    int off;
    int n;
    XQ_HD void put(int from, int to)
    {
        if (EMIT) {
            const int i = off + n;
            if (i < MAXMOVES) {
                lab[i] = label_of(from, to);
                ft[i] = (uint16_t)((from << 8) | to);
            }
        }
        ++n;
    }
};

// Moves of the piece standing on square s (0 if empty / opponent), reference order.
template <bool EMIT>
XQ_HD int gen_sq(const int8_t* b, int s, uint16_t* lab, uint16_t* ft, int off)
{
    const int p = b[s];
    if (p <= 0) return 0;
    MoveSink<EMIT> out{lab, ft, off, 0};
    const int x = s % 9, y = s / 9;
    if (p == ROOK || p == CANNON) {                       // static_env.py:288-320
        int l = x - 1, r = x + 1, d = y - 1, u = y + 1;
        while (l > -1 && b[y * 9 + l] == 0) --l;
        while (r < 9 && b[y * 9 + r] == 0) ++r;
        while (d > -1 && b[d * 9 + x] == 0) --d;
        while (u < 10 && b[u * 9 + x] == 0) ++u;
        for (int t = l + 1; t < x; ++t) out.put(s, y * 9 + t);
        for (int t = x + 1; t < r; ++t) out.put(s, y * 9 + t);
        for (int t = d + 1; t < y; ++t) out.put(s, t * 9 + x);
        for (int t = y + 1; t < u; ++t) out.put(s, t * 9 + x);
        if (p == CANNON) {                                // jump to the next blocker beyond each screen
            if (l > -1) { --l; while (l > -1 && b[y * 9 + l] == 0) --l; }
            if (r < 9) { ++r; while (r < 9 && b[y * 9 + r] == 0) ++r; }
            if (d > -1) { --d; while (d > -1 && b[d * 9 + x] == 0) --d; }
            if (u < 10) { ++u; while (u < 10 && b[u * 9 + x] == 0) ++u; }
        }
        // a blocker is never empty, so can_move() == "on board and not the mover's own piece"
        if (l > -1 && b[y * 9 + l] < 0) out.put(s, y * 9 + l);
        if (r < 9 && b[y * 9 + r] < 0) out.put(s, y * 9 + r);
        if (d > -1 && b[d * 9 + x] < 0) out.put(s, d * 9 + x);
        if (u < 10 && b[u * 9 + x] < 0) out.put(s, u * 9 + x);
        return out.n;
    }
    if (p == KING) {                                      // :277-286
        const int dx[4] = {0, 1, 0, -1}, dy[4] = {-1, 0, 1, 0};
        int u = y + 1;
        while (u < 10 && b[u * 9 + x] == 0) ++u;
        const bool flying = (u < 10 && b[u * 9 + x] == -KING);
        for (int k = 0; k < 4; ++k) {
            const int x_ = x + dx[k], y_ = y + dy[k];
            if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9 || b[y_ * 9 + x_] > 0) continue;
            if (x_ < 3 || x_ > 5 || y_ > 2) continue;
            out.put(s, y_ * 9 + x_);
            if (flying) out.put(s, u * 9 + x);            // emitted once per accepted king step
        }
        return out.n;
    }
    if (p == ADVISOR) {
        const int dx[4] = {-1, 1, -1, 1}, dy[4] = {-1, -1, 1, 1};
        for (int k = 0; k < 4; ++k) {
            const int x_ = x + dx[k], y_ = y + dy[k];
            if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9 || b[y_ * 9 + x_] > 0) continue;
            if (x_ < 3 || x_ > 5 || y_ > 2) continue;
            out.put(s, y_ * 9 + x_);
        }
        return out.n;
    }
    if (p == ELEPHANT) {                                  // :272-276
        const int dx[4] = {-2, 2, 2, -2}, dy[4] = {-2, -2, 2, 2};
        for (int k = 0; k < 4; ++k) {
            const int x_ = x + dx[k], y_ = y + dy[k];
            if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9 || b[y_ * 9 + x_] > 0) continue;
            if (b[(y + dy[k] / 2) * 9 + x + dx[k] / 2] != 0) continue;
            if (y_ > 4) continue;
            out.put(s, y_ * 9 + x_);
        }
        return out.n;
    }
    if (p == KNIGHT) {
        const int dx[8] = {-1, 1, 2, 2, 1, -1, -2, -2}, dy[8] = {-2, -2, -1, 1, 2, 2, 1, -1};
        for (int k = 0; k < 8; ++k) {
            const int x_ = x + dx[k], y_ = y + dy[k];
            if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9 || b[y_ * 9 + x_] > 0) continue;
            if (b[(y + dy[k] / 2) * 9 + x + dx[k] / 2] != 0) continue;   // int(d/2) truncates toward 0
            out.put(s, y_ * 9 + x_);
        }
        return out.n;
    }
    if (p == PAWN) {                                      // :270
        const int dx[3] = {0, -1, 1}, dy[3] = {1, 0, 0};
        for (int k = 0; k < 3; ++k) {
            const int x_ = x + dx[k], y_ = y + dy[k];
            if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9 || b[y_ * 9 + x_] > 0) continue;
            if (y < 5 && x_ != x) continue;
            out.put(s, y_ * 9 + x_);
        }
        return out.n;
    }
    return 0;
}

// Value (0/1) of input-plane element o = c*90 + i*9 + j for a board (static_env.py:137-156):
// channel = type-1 for the mover, 7 + type-1 for the opponent; row i of the planes is y = 9 - i.
XQ_HD int plane_bit(const int8_t* b, int o)
{
    const int c = o / 90, rem = o - c * 90;
    const int i = rem / 9, j = rem - i * 9;
    const int p = b[(9 - i) * 9 + j];
    const int want = (c < 7) ? (c + 1) : -(c - 7 + 1);
    return p == want;
}

// 4-bit packing of a board: 0 empty, 1..7 mover, 9..15 opponent.  90 nibbles -> 12 u32 (last word padded).
XQ_HD uint32_t nib_of(int p) { return (uint32_t)(p >= 0 ? p : (8 - p)); }
XQ_HD int piece_of_nib(uint32_t v) { return v < 8 ? (int)v : 8 - (int)v; }

// 64-bit mix (splitmix64 finaliser) used for the transposition hash.
XQ_HD uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

}  // namespace xq
