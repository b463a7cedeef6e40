// xq_tpb.h -- the Xiangqi rules for ONE board executed by ONE thread (host + device).
//
// The batch kernels (cz_movegen / cz_done / cz_rules_fused on thousands to millions of independent boards)
// run one board per LANE: 64 boards per wavefront, every lane busy, boards and move lists staged in LDS so that
// the global loads and stores stay coalesced.  (The search kernel keeps one wavefront per game: there the
// parallelism is inside a tree.)  Same semantics and move order as static_env.py, built from the same
// per-piece generator (xq_lane.h::gen_piece) as the wave-cooperative rules.
#pragma once
#include "xq_lane.h"

namespace xq {

XQ_HD uint64_t rev64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(v);
#else
    uint64_t r = 0;
    for (int i = 0; i < 64; ++i) r |= ((v >> i) & 1ull) << (63 - i);
    return r;
#endif
}
// square set seen from the other side: bit s -> bit 89 - s (fliped_state, static_env.py:245-254)
XQ_HD Set90 flip_set(const Set90& m)
{
    const uint64_t lo = rev64(m.hi), hi = rev64(m.lo);       // bit s -> bit 127 - s
    Set90 r;
    r.lo = (lo >> 38) | (hi << 26);                           // then down by 38
    r.hi = hi >> 38;
    return r;
}
XQ_HD int first_sq(const Set90& m)                            // lowest set square, -1 if empty
{
    if (m.lo) return __builtin_ctzll(m.lo);
    if (m.hi) return 64 + __builtin_ctzll(m.hi);
    return -1;
}
XQ_HD int last_sq(const Set90& m)
{
    if (m.hi) return 64 + 63 - __builtin_clzll(m.hi);
    if (m.lo) return 63 - __builtin_clzll(m.lo);
    return -1;
}
XQ_HD void set_sq(Set90& m, int s)
{
    if (s < 64) m.lo |= 1ull << s; else m.hi |= 1ull << (s - 64);
}

struct BoardSets {
    Set90 occ, own, oking, mking;     // all pieces, mover's pieces, opponent's king(s), mover's king(s)
};

XQ_HD BoardSets board_sets(const int8_t* b)
{
    BoardSets t{{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int s = 0; s < NSQ; ++s) {
        const int p = b[s];
        if (p == 0) continue;
        set_sq(t.occ, s);
        if (p > 0) set_sq(t.own, s);
        if (p == -KING) set_sq(t.oking, s);
        if (p == KING) set_sq(t.mking, s);
    }
    return t;
}

// get_legal_moves (static_env.py:256-321) into lab[0..); returns the count.  *hit = index of the first move
// landing on `watch` (-1: none).
XQ_HD int tpb_movegen(const int8_t* b, const BoardSets& t, uint16_t* lab, int watch, int* hit)
{
    int n = 0;
    Set90 rest = t.own;
    for (;;) {
        const int s = first_sq(rest);
        if (s < 0) break;
        if (s < 64) rest.lo &= rest.lo - 1; else rest.hi &= rest.hi - 1;
        n += gen_piece<true>(b[s], s, t.occ, t.own, t.oking, lab, nullptr, n, watch, hit);   // table labels: measured faster
    }
    return n;
}

// is square `target` (opponent's frame) attacked by the opponent, i.e. does get_legal_moves(fliped_state)
// contain a move landing on it (static_env.py:61-70)
XQ_HD bool tpb_opponent_reaches(const int8_t* b, const BoardSets& t, int target)
{
    Set90 opp{t.occ.lo & ~t.own.lo, t.occ.hi & ~t.own.hi};
    const Set90 occ_r = flip_set(t.occ), own_r = flip_set(opp), oking_r = flip_set(t.mking);
    Set90 rest = own_r;
    int hit = -1;
    for (;;) {
        const int s = first_sq(rest);
        if (s < 0) break;
        if (s < 64) rest.lo &= rest.lo - 1; else rest.hi &= rest.hi - 1;
        gen_piece<false>(-b[89 - s], s, occ_r, own_r, oking_r, nullptr, nullptr, 0, target, &hit);
        if (hit >= 0) return true;
    }
    return false;
}

struct TpbResult {
    int n;            // move count (list in lab[]); always generated, also for positions decided early
    int over, v, final_move, check;
};

// done(state, need_check) + the position's move list (static_env.py:14-77)
XQ_HD TpbResult tpb_rules(const int8_t* b, uint16_t* lab, bool need_check)
{
    TpbResult r{0, 0, 0, NOMOVE, 0};
    const BoardSets t = board_sets(b);
    const int rk = last_sq(t.mking), bk = last_sq(t.oking);      // scan order: the last king found wins
    int hit = -1;
    r.n = tpb_movegen(b, t, lab, bk, &hit);
    if (bk < 0) { r.over = 1; r.v = 1; return r; }               // 's' not in state
    if (rk < 0) { r.over = 1; r.v = -1; return r; }              // 'S' not in state
    const int rx = rk % 9, ry = rk / 9, bx = bk % 9, by = bk / 9;
    int winner = 0;
    if (ry == 0 && rx == 0) { winner = 2; r.v = -1; }            // dead branches kept, :33-38
    else if (by == 0 && bx == 0) { winner = 1; r.v = 1; }
    else if (rx == bx) {                                         // kings on one file, :39-49
        bool blocked = false;
        for (int y = ry + 1; y < by; ++y) blocked = blocked || has(t.occ, y * 9 + rx);
        if (!blocked) { r.v = 1; winner = 1; }
    }
    if (!winner && hit >= 0) { winner = 1; r.v = 1; r.final_move = lab[hit < MAXMOVES ? hit : 0]; }   // :52-60
    if (!winner && need_check) r.check = tpb_opponent_reaches(b, t, 89 - rk) ? 1 : 0;              // :61-73
    r.over = winner != 0;
    return r;
}

}  // namespace xq
