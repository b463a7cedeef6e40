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

// labels by arithmetic from the per-square block (xq_lane.h::label_in_block): 3.28 TB/s on the 1 M-board micro-suite
// against 3.07 with the (from, to) table gather; both forms are checked against the oracle on the CPU (test_lane_cpu.py)
constexpr bool TPB_FORMULA_LABELS = true;

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

// ---- divergence-free ordering of the per-lane work ---------------------------------------------------------
// 64 lanes work on 64 different boards.  Walking each board's pieces in square order makes every lane meet a
// different piece type in the same loop iteration, so the wave executes the union of all per-type code paths every
// time.  Instead each lane first sorts its pieces by TYPE (packed square lists), and the move generator runs one
// loop per type with the type a compile-time constant: all lanes execute the same specialised code.  The
// reference's list order (by square) is restored with per-piece offsets: pass 1 counts, a prefix over the pieces in
// square order gives each piece its offset, pass 2 emits there.
struct TypeLists {
    uint64_t sq[8];     // per type: up to 9 squares, 7 bits each (a side has at most 5 pieces of one type)
    uint8_t n[8];
    bool ok;            // false: more pieces of one type than the lists hold -> caller uses the generic path
};

XQ_HD TypeLists type_lists(const int8_t* b, Set90 side, bool negate)
{
    TypeLists t;
    for (int k = 0; k < 8; ++k) { t.sq[k] = 0; t.n[k] = 0; }
    t.ok = true;
    for (;;) {
        const int s = first_sq(side);
        if (s < 0) break;
        if (s < 64) side.lo &= side.lo - 1; else side.hi &= side.hi - 1;
        const int p = negate ? -b[s] : b[s];
        // predicated appends, written out per type: no divergent control flow and no dynamic register indexing
#define XQ_TL(k)                                                                           \
        {                                                                                      \
            const bool hit_ = p == (k);                                                        \
            if (hit_ && t.n[k] >= 9) t.ok = false;                                             \
            if (hit_ && t.n[k] < 9) { t.sq[k] |= (uint64_t)s << (7 * t.n[k]); t.n[k] += 1; }  \
        }
        XQ_TL(1) XQ_TL(2) XQ_TL(3) XQ_TL(4) XQ_TL(5) XQ_TL(6) XQ_TL(7)
#undef XQ_TL
    }
    return t;
}

struct PieceCounts {        // per piece (indexed by its rank in square order): move count, then list offset
    uint64_t c[3];          // 8 bits x 24 pieces
    XQ_HD void set(int r, uint32_t v)
    {
        const uint64_t m = (uint64_t)v << (8 * (r & 7));
        if (r < 8) c[0] |= m; else if (r < 16) c[1] |= m; else c[2] |= m;
    }
    XQ_HD uint32_t get(int r) const
    {
        const uint64_t w = r < 8 ? c[0] : (r < 16 ? c[1] : c[2]);
        return (uint32_t)(w >> (8 * (r & 7))) & 0xFFu;
    }
};

XQ_HD int rank_of(const Set90& own, int s)
{
    if (s < 64) return __builtin_popcountll(own.lo & ((1ull << s) - 1ull));
    return __builtin_popcountll(own.lo) + __builtin_popcountll(own.hi & ((1ull << (s - 64)) - 1ull));
}

// one type, all of this lane's pieces of that type.  COUNT: fill pc with the move counts; otherwise emit at pc's offsets
template <int TYPE, bool COUNT>
XQ_HD void tpb_type_pass(const TypeLists& tl, const BoardSets& t, PieceCounts& pc, uint16_t* lab, int watch, int* hit,
                         int cap, int* hit_from)
{
    uint64_t list = tl.sq[TYPE];
    for (int i = 0; i < tl.n[TYPE]; ++i) {
        const int s = (int)(list & 0x7F);
        list >>= 7;
        const int r = rank_of(t.own, s);
        if (COUNT) {
            pc.set(r, (uint32_t)gen_piece<false>(TYPE, s, t.occ, t.own, t.oking, nullptr, nullptr, 0));
        } else {
            int h = -1, hf = -1;
            gen_piece<true>(TYPE, s, t.occ, t.own, t.oking, lab, nullptr, (int)pc.get(r), watch, &h, TPB_FORMULA_LABELS, cap, &hf);
            if (h >= 0 && (*hit < 0 || h < *hit)) { *hit = h; *hit_from = hf; }   // first in LIST order, not in type order
        }
    }
}

template <bool COUNT>
XQ_HD void tpb_all_types(const TypeLists& tl, const BoardSets& t, PieceCounts& pc, uint16_t* lab, int watch, int* hit,
                         int cap, int* hit_from)
{
    tpb_type_pass<ROOK, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
    tpb_type_pass<CANNON, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
    tpb_type_pass<KNIGHT, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
    tpb_type_pass<PAWN, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
    tpb_type_pass<ELEPHANT, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
    tpb_type_pass<ADVISOR, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
    tpb_type_pass<KING, COUNT>(tl, t, pc, lab, watch, hit, cap, hit_from);
}

// generic (square-ordered) generator: any board, any number of pieces
XQ_HD int tpb_movegen_generic(const int8_t* b, const BoardSets& t, uint16_t* lab, int watch, int* hit, int cap,
                              int* hit_from)
{
    int n = 0;
    Set90 rest = t.own;
    for (;;) {
        const int s = first_sq(rest);
        if (s < 0) break;
        if (s < 64) rest.lo &= rest.lo - 1; else rest.hi &= rest.hi - 1;
        n += gen_piece<true>(b[s], s, t.occ, t.own, t.oking, lab, nullptr, n, watch, hit, TPB_FORMULA_LABELS, cap, hit_from);
    }
    return n;
}

// get_legal_moves (static_env.py:256-321) into lab[0..cap); returns the count (moves beyond cap are counted, not
// stored).  *hit = index of the first move landing on `watch` (-1: none), *hit_from its source square.
XQ_HD int tpb_movegen(const int8_t* b, const BoardSets& t, uint16_t* lab, int watch, int* hit, int cap, int* hit_from)
{
    const int np = __builtin_popcountll(t.own.lo) + __builtin_popcountll(t.own.hi);
    const TypeLists tl = type_lists(b, t.own, false);
    if (np > 24 || !tl.ok) return tpb_movegen_generic(b, t, lab, watch, hit, cap, hit_from);
    PieceCounts pc{{0, 0, 0}};
    tpb_all_types<true>(tl, t, pc, nullptr, -1, nullptr, cap, nullptr);
    int total = 0;
    PieceCounts off{{0, 0, 0}};
    for (int r = 0; r < np; ++r) {                          // offsets in square order
        const uint32_t c = pc.get(r);
        off.set(r, (uint32_t)(total < 255 ? total : 255));
        total += (int)c;
    }
    if (total > 255) return tpb_movegen_generic(b, t, lab, watch, hit, cap, hit_from);     // impossible boards only
    tpb_all_types<false>(tl, t, off, lab, watch, hit, cap, hit_from);
    return total;
}

template <int TYPE>
XQ_HD void reach_type(const TypeLists& tl, const BoardSets& f, int target, int* hit)
{
    uint64_t list = tl.sq[TYPE];
    for (int i = 0; i < tl.n[TYPE]; ++i) {
        const int s = 89 - (int)(list & 0x7F);              // the lists hold squares of OUR frame
        list >>= 7;
        gen_piece<false>(TYPE, s, f.occ, f.own, f.oking, nullptr, nullptr, 0, target, hit);
    }
}

// is square `target` (opponent's frame) attacked by the opponent, i.e. does get_legal_moves(fliped_state)
// contain a move landing on it (static_env.py:61-70)
XQ_HD bool tpb_opponent_reaches(const int8_t* b, const BoardSets& t, int target)
{
    const Set90 opp{t.occ.lo & ~t.own.lo, t.occ.hi & ~t.own.hi};
    BoardSets f;                                            // the position seen by the opponent
    f.occ = flip_set(t.occ); f.own = flip_set(opp); f.oking = flip_set(t.mking); f.mking = flip_set(t.oking);
    const TypeLists tl = type_lists(b, opp, true);          // squares still in OUR frame
    int hit = -1;
    if (tl.ok) {
        // only "does any move land on target" matters: no ordering, no offsets; one specialised loop per type
        reach_type<ROOK>(tl, f, target, &hit);
        reach_type<CANNON>(tl, f, target, &hit);
        reach_type<KNIGHT>(tl, f, target, &hit);
        reach_type<PAWN>(tl, f, target, &hit);
        reach_type<ELEPHANT>(tl, f, target, &hit);
        reach_type<ADVISOR>(tl, f, target, &hit);
        reach_type<KING>(tl, f, target, &hit);
        return hit >= 0;
    }
    Set90 rest = f.own;
    for (;;) {
        const int s = first_sq(rest);
        if (s < 0) break;
        if (s < 64) rest.lo &= rest.lo - 1; else rest.hi &= rest.hi - 1;
        gen_piece<false>(-b[89 - s], s, f.occ, f.own, f.oking, nullptr, nullptr, 0, target, &hit);
        if (hit >= 0) return true;
    }
    return false;
}

struct TpbResult {
    int n;            // move count (list in lab[]); always generated, also for positions decided early
    int over, v, final_move, check;
};

// done(state, need_check) + the position's move list (static_env.py:14-77)
XQ_HD TpbResult tpb_rules(const int8_t* b, uint16_t* lab, bool need_check, int cap = MAXMOVES)
{
    TpbResult r{0, 0, 0, NOMOVE, 0};
    const BoardSets t = board_sets(b);
    const int rk = last_sq(t.mking), bk = last_sq(t.oking);      // scan order: the last king found wins
    int hit = -1, hit_from = -1;
    r.n = tpb_movegen(b, t, lab, bk, &hit, cap, &hit_from);
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
    if (!winner && hit >= 0) { winner = 1; r.v = 1; r.final_move = label_of(hit_from, bk); }         // :52-60
    if (!winner && need_check) r.check = tpb_opponent_reaches(b, t, 89 - rk) ? 1 : 0;              // :61-73
    r.over = winner != 0;
    return r;
}

}  // namespace xq
