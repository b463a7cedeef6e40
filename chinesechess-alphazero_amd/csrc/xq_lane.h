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

// Step tables of the non-sliding pieces in the reference's direction order (mov_dir, common.py:66-76):
// packed as (dx + 2) | (dy + 2) << 3.  Row = piece code (KNIGHT, ELEPHANT, ADVISOR, KING, PAWN).
#define XQ_STEP(dx, dy) (uint8_t)(((dx) + 2) | (((dy) + 2) << 3))
struct StepTable {
    uint8_t n[8];
    uint8_t d[8][8];
};
constexpr StepTable make_steps()
{
    StepTable t{};
    const uint8_t king[4] = {XQ_STEP(0, -1), XQ_STEP(1, 0), XQ_STEP(0, 1), XQ_STEP(-1, 0)};
    const uint8_t adv[4] = {XQ_STEP(-1, -1), XQ_STEP(1, -1), XQ_STEP(-1, 1), XQ_STEP(1, 1)};
    const uint8_t ele[4] = {XQ_STEP(-2, -2), XQ_STEP(2, -2), XQ_STEP(2, 2), XQ_STEP(-2, 2)};
    const uint8_t kn[8] = {XQ_STEP(-1, -2), XQ_STEP(1, -2), XQ_STEP(2, -1), XQ_STEP(2, 1),
                           XQ_STEP(1, 2), XQ_STEP(-1, 2), XQ_STEP(-2, 1), XQ_STEP(-2, -1)};
    const uint8_t pawn[3] = {XQ_STEP(0, 1), XQ_STEP(-1, 0), XQ_STEP(1, 0)};
    for (int i = 0; i < 8; ++i) { t.n[i] = 0; for (int k = 0; k < 8; ++k) t.d[i][k] = 0; }
    t.n[KING] = 4; t.n[ADVISOR] = 4; t.n[ELEPHANT] = 4; t.n[KNIGHT] = 8; t.n[PAWN] = 3;
    for (int k = 0; k < 4; ++k) { t.d[KING][k] = king[k]; t.d[ADVISOR][k] = adv[k]; t.d[ELEPHANT][k] = ele[k]; }
    for (int k = 0; k < 8; ++k) t.d[KNIGHT][k] = kn[k];
    for (int k = 0; k < 3; ++k) t.d[PAWN][k] = pawn[k];
    return t;
}
#if defined(__HIPCC__)
static __device__ const StepTable d_steps = make_steps();
#endif
static constexpr StepTable h_steps = make_steps();

XQ_HD int step_count(int p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return d_steps.n[p];
#else
    return h_steps.n[p];
#endif
}
XQ_HD int step_code(int p, int k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return d_steps.d[p][k];
#else
    return h_steps.d[p][k];
#endif
}

// Moves of the piece standing on square s (0 if empty / opponent), reference order (static_env.py:256-321).
template <bool EMIT>
XQ_HD int gen_sq(const int8_t* b, int s, uint16_t* lab, uint16_t* ft, int off)
{
    const int p = b[s];
    if (p <= 0) return 0;
    MoveSink<EMIT> out{lab, ft, off, 0};
    const int x = s % 9, y = s / 9;
    if (p == ROOK || p == CANNON) {                       // :288-320
        // nearest blockers on the four rays (x_board_from / y_board_from, :332-348); sentinels -1 / 9 / 10
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
    // stepping pieces: one table-driven loop (:264-286)
    int fly = -1;                                         // flying-king capture target, -1 if none
    if (p == KING) {
        int u = y + 1;
        while (u < 10 && b[u * 9 + x] == 0) ++u;
        if (u < 10 && b[u * 9 + x] == -KING) fly = u * 9 + x;
    }
    const int nd = step_count(p);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int k = 0; k < nd; ++k) {
        const int code = step_code(p, k);
        const int dx = (code & 7) - 2, dy = (code >> 3) - 2;
        const int x_ = x + dx, y_ = y + dy;
        if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9) continue;           // can_move, :323-330
        const int t = y_ * 9 + x_;
        if (b[t] > 0) continue;
        if (p == PAWN) {
            if (y < 5 && x_ != x) continue;                           // :270
        } else if (p == KNIGHT || p == ELEPHANT) {
            if (b[(y + dy / 2) * 9 + x + dx / 2] != 0) continue;      // leg / eye; int(d/2) truncates toward 0
            if (p == ELEPHANT && y_ > 4) continue;                    // :275
        } else {                                                      // king, advisor: palace (:277-281)
            if (x_ < 3 || x_ > 5 || y_ > 2) continue;
        }
        out.put(s, t);
        if (fly >= 0) out.put(s, fly);                                // once per accepted king step (:283-286)
    }
    return out.n;
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
