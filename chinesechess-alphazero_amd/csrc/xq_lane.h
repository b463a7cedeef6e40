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

// Label of a rook / cannon / king / pawn / knight move without the 16 KB (from, to) table: the label set
// lists, per source square, the 8 same-rank and 9 same-file destinations and then the on-board knight jumps
// (lookup_tables.py:66-77), so the index is base[from] + a position computed from the coordinates.
// Advisor and elephant moves live in the literal tail of the set and keep using the table.
// base = Tables::base[from], valid = Tables::kvalid[from]: the same for every move of a piece, so a caller that emits
// several moves from one square loads them once (label_block) instead of once per move -- each is a dependent global
// load, and in the one-wave-per-tree search they were most of the move generator's time.
XQ_HD uint16_t label_in_block(int base, uint32_t valid, int from, int to)
{
    const int x = from % 9, y = from / 9, tx = to % 9, ty = to / 9;
    if (ty == y) return (uint16_t)(base + (tx < x ? tx : tx - 1));
    if (tx == x) return (uint16_t)(base + 8 + (ty < y ? ty : ty - 1));
    // knight: offsets (dy, dx) in label order {-2,-1},{-1,-2},{-2,1},{1,-2},{2,-1},{-1,2},{2,1},{1,2}
    const int dy = ty - y, dx = tx - x;
    int k;
    if (dy == -2) k = dx < 0 ? 0 : 2;
    else if (dy == 2) k = dx < 0 ? 4 : 6;
    else if (dy == -1) k = dx < 0 ? 1 : 5;
    else k = dx < 0 ? 3 : 7;
    return (uint16_t)(base + 17 + __builtin_popcount(valid & ((1u << k) - 1u)));
}
XQ_HD void label_block(int from, int* base, uint32_t* valid)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *base = d_tab.base[from]; *valid = d_tab.kvalid[from];
#else
    *base = h_tab.base[from]; *valid = h_tab.kvalid[from];
#endif
}
XQ_HD uint16_t label_of_line_or_knight(int from, int to)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const Tables& T = d_tab;
#else
    const Tables& T = h_tab;
#endif
    const int x = from % 9, y = from / 9, tx = to % 9, ty = to / 9;
    const int base = T.base[from];
    if (ty == y) return (uint16_t)(base + (tx < x ? tx : tx - 1));
    if (tx == x) return (uint16_t)(base + 8 + (ty < y ? ty : ty - 1));
    // knight: offsets (dy, dx) in label order {-2,-1},{-1,-2},{-2,1},{1,-2},{2,-1},{-1,2},{2,1},{1,2}
    const int dy = ty - y, dx = tx - x;
    int k;
    if (dy == -2) k = dx < 0 ? 0 : 2;
    else if (dy == 2) k = dx < 0 ? 4 : 6;
    else if (dy == -1) k = dx < 0 ? 1 : 5;
    else k = dx < 0 ? 3 : 7;
    const uint32_t valid = T.kvalid[from];
    return (uint16_t)(base + 17 + __builtin_popcount(valid & ((1u << k) - 1u)));
}

// Sink for generated moves: an ordered list of (label, from<<8|to) starting at `off`.
// EMIT=false only counts (first pass of the two-pass ordered compaction).
template <bool EMIT>
struct MoveSink {
    uint16_t* lab;
    uint16_t* ft;               // may be null
    int off;
    int n;
    int cap;                    // entries the list can hold (moves beyond it are counted but not stored)
    int watch;                  // destination square to look for (-1: none)
    int hit;                    // list index of the first move landing on `watch`, -1 if none
    int hit_from;               // its source square
    bool formula;               // label by arithmetic instead of the 16 KB table (one board per lane: the table
                                // gather is a dependent global load per move and there are few waves to hide it)
    int base;                   // formula: the source square's label block (label_block), loaded once per piece
    uint32_t kvalid;
    XQ_HD void put(int from, int to)
    {
        if (to == watch && hit < 0) { hit = off + n; hit_from = from; }
        if (EMIT) {
            const int i = off + n;
            if (i < cap) {
                lab[i] = formula ? label_in_block(base, kvalid, from, to) : label_of(from, to);
                if (ft) ft[i] = (uint16_t)((from << 8) | to);
            }
        }
        ++n;
    }
    // a rook / cannon move along a rank or a file: in formula mode its label is base + a coordinate the caller has at hand
    XQ_HD void put_line(int from, int to, int formula_label)
    {
        if (to == watch && hit < 0) { hit = off + n; hit_from = from; }
        if (EMIT) {
            const int i = off + n;
            if (i < cap) {
                lab[i] = formula ? (uint16_t)formula_label : label_of(from, to);
                if (ft) ft[i] = (uint16_t)((from << 8) | to);
            }
        }
        ++n;
    }
    XQ_HD void put_labelled(int from, int to, uint16_t label)      // the caller already has the label (AeLabels)
    {
        if (to == watch && hit < 0) { hit = off + n; hit_from = from; }
        if (EMIT) {
            const int i = off + n;
            if (i < cap) {
                lab[i] = label;
                if (ft) ft[i] = (uint16_t)((from << 8) | to);
            }
        }
        ++n;
    }
};

// Step tables of the non-sliding pieces in the reference's direction order (mov_dir, common.py:66-76):
// packed as (dx + 2) | (dy + 2) << 3.  Row = piece code (KNIGHT, ELEPHANT, ADVISOR, KING, PAWN).
#define XQ_STEP(dx, dy) (uint8_t)(((dx) + 2) | (((dy) + 2) << 3))
struct alignas(8) StepTable {
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
// all (up to 8) step codes of piece p, code k in byte k, and their number, as immediates selected by the piece type:
// no memory access (a one-byte table load per loop iteration was a dependent global round trip each)
constexpr uint64_t step_row_of(int p)
{
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v |= (uint64_t)h_steps.d[p][k] << (8 * k);
    return v;
}
XQ_HD uint64_t step_row(int p)
{
    constexpr uint64_t RK = step_row_of(KING), RA = step_row_of(ADVISOR), RE = step_row_of(ELEPHANT),
                       RN = step_row_of(KNIGHT), RP = step_row_of(PAWN);
    return p == KNIGHT ? RN : (p == PAWN ? RP : (p == KING ? RK : (p == ADVISOR ? RA : (p == ELEPHANT ? RE : 0ull))));
}
XQ_HD int step_count_imm(int p)
{
    return p == KNIGHT ? 8 : (p == PAWN ? 3 : ((p == KING || p == ADVISOR || p == ELEPHANT) ? 4 : 0));
}
static_assert(h_steps.n[KNIGHT] == 8 && h_steps.n[PAWN] == 3 && h_steps.n[KING] == 4 && h_steps.n[ADVISOR] == 4 &&
              h_steps.n[ELEPHANT] == 4 && h_steps.n[ROOK] == 0 && h_steps.n[CANNON] == 0, "step_count_imm");

// Labels of the (up to 4) advisor / elephant moves from every square, step k of the piece's row in 16-bit field k
// (NOMOVE where the step leaves the board or is not in the label set): one 8-byte load per piece instead of a
// (from, to) table gather per move.
struct alignas(8) AeLabels {
    uint16_t lab[2][NSQ][4];    // [0] advisor, [1] elephant
};
constexpr AeLabels make_ae_labels()
{
    AeLabels t{};
    for (int e = 0; e < 2; ++e)
        for (int sq = 0; sq < NSQ; ++sq)
            for (int k = 0; k < 4; ++k) {
                const int code = h_steps.d[e ? ELEPHANT : ADVISOR][k];
                const int x_ = sq % 9 + (code & 7) - 2, y_ = sq / 9 + (code >> 3) - 2;
                t.lab[e][sq][k] = (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9) ? (uint16_t)NOMOVE
                                                                        : h_tab.label_of[sq * NSQ + y_ * 9 + x_];
            }
    return t;
}
#if defined(__HIPCC__)
static __device__ const AeLabels d_ae = make_ae_labels();
#endif
static constexpr AeLabels h_ae = make_ae_labels();
XQ_HD uint64_t ae_label_row(int elephant, int sq)
{
    uint64_t v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    v = *reinterpret_cast<const uint64_t*>(d_ae.lab[elephant][sq]);
#else
    for (int k = 0; k < 4; ++k) v |= (uint64_t)h_ae.lab[elephant][sq][k] << (16 * k);
#endif
    return v;
}

// 90-bit square sets (bit s = square s): occupancy, the mover's pieces, the opponent's king(s).
struct Set90 {
    uint64_t lo, hi;
};
XQ_HD bool has(const Set90& m, int s) { return s < 64 ? (m.lo >> s) & 1ull : (m.hi >> (s - 64)) & 1ull; }
// the 9 squares of rank y as bits 0..8
XQ_HD uint32_t rank_bits(const Set90& m, int y)
{
    const int pos = 9 * y;
    uint64_t v;
    if (pos >= 64) v = m.hi >> (pos - 64);
    else if (pos + 9 <= 64) v = m.lo >> pos;
    else v = (m.lo >> pos) | (m.hi << (64 - pos));
    return (uint32_t)v & 0x1FFu;
}
// the 10 squares of file x as bits 0..9
XQ_HD uint32_t file_bits(const Set90& m, int x)
{
    // ranks 0..6 (squares x + 9k <= 62, all in `lo`): mask the seven bits 9 apart and gather them with one multiply --
    // bit 9k times 2^(48 - 8j) lands on 48 + k for j = k and on 48 + k + 8 (k - j) otherwise, all distinct, so nothing
    // carries into bits 48..54.  (A loop of ten has() calls was ~60 instructions; both slider lanes and the king use this.)
    const uint64_t seven = ((m.lo >> x) & 0x0040201008040201ULL) * 0x0001010101010101ULL;
    uint32_t v = (uint32_t)(seven >> 48) & 0x7Fu;
    const uint64_t from63 = (m.lo >> 63) | (m.hi << 1);               // bit i = square 63 + i
    v |= (uint32_t)((from63 >> x) & 1ull) << 7;                       // rank 7: square 63 + x
    v |= (uint32_t)((m.hi >> (x + 8)) & 1ull) << 8;                   // rank 8: square 72 + x
    v |= (uint32_t)((m.hi >> (x + 17)) & 1ull) << 9;                  // rank 9: square 81 + x
    return v;
}
XQ_HD int top_bit(uint32_t v) { return 31 - __builtin_clz(v); }       // v != 0
XQ_HD int low_bit(uint32_t v) { return __builtin_ctz(v); }            // v != 0

// Moves of the mover's piece `p` (> 0) standing on square s, reference order (static_env.py:256-321).
// The board enters only through three square sets, so no lane touches LDS while generating:
//   occ = all pieces, own = the mover's pieces, oking = the opponent's king(s).
template <bool EMIT>
XQ_HD int gen_piece(int p, int s, const Set90& occ, const Set90& own, const Set90& oking,
                    uint16_t* lab, uint16_t* ft, int off, int watch = -1, int* hit = nullptr,
                    bool formula_labels = false, int cap = MAXMOVES, int* hit_from = nullptr)
{
    // advisor / elephant moves live in the literal tail of the label set: always from the table
    MoveSink<EMIT> out{lab, ft, off, 0, cap, watch, -1, -1, formula_labels && p != ADVISOR && p != ELEPHANT, 0, 0u};
    if (EMIT && out.formula) label_block(s, &out.base, &out.kvalid);
    const int x = s % 9, y = s / 9;
    if (p == ROOK || p == CANNON) {                       // :288-320
        const uint32_t row = rank_bits(occ, y), col = file_bits(occ, x);
        // nearest blockers on the four rays (x_board_from / y_board_from, :332-348); sentinels -1 / 9 / 10
        uint32_t lm = row & ((1u << x) - 1u), rm = row >> (x + 1);
        uint32_t dm = col & ((1u << y) - 1u), um = col >> (y + 1);
        int l = lm ? top_bit(lm) : -1, r = rm ? x + 1 + low_bit(rm) : 9;
        int d = dm ? top_bit(dm) : -1, u = um ? y + 1 + low_bit(um) : 10;
        // labels of the block of square s: rank destinations tx -> base + (tx < x ? tx : tx - 1), file destinations
        // ty -> base + 8 + (ty < y ? ty : ty - 1)   (label_in_block)
        const int b0 = out.base, b8 = out.base + 8;
        if (!EMIT && watch < 0) {
            out.n += (x - l - 1) + (r - x - 1) + (y - d - 1) + (u - y - 1);      // counting only: no loop
        } else {
            for (int t = l + 1; t < x; ++t) out.put_line(s, y * 9 + t, b0 + t);
            for (int t = x + 1; t < r; ++t) out.put_line(s, y * 9 + t, b0 + t - 1);
            for (int t = d + 1; t < y; ++t) out.put_line(s, t * 9 + x, b8 + t);
            for (int t = y + 1; t < u; ++t) out.put_line(s, t * 9 + x, b8 + t - 1);
        }
        if (p == CANNON) {                                // the next blocker beyond each screen
            if (l > -1) { lm &= ~(1u << l); l = lm ? top_bit(lm) : -1; }
            if (r < 9) { rm &= rm - 1u; r = rm ? x + 1 + low_bit(rm) : 9; }
            if (d > -1) { dm &= ~(1u << d); d = dm ? top_bit(dm) : -1; }
            if (u < 10) { um &= um - 1u; u = um ? y + 1 + low_bit(um) : 10; }
        }
        // a blocker is never empty, so can_move() == "on board and not the mover's own piece"
        if (l > -1 && !has(own, y * 9 + l)) out.put_line(s, y * 9 + l, b0 + l);
        if (r < 9 && !has(own, y * 9 + r)) out.put_line(s, y * 9 + r, b0 + r - 1);
        if (d > -1 && !has(own, d * 9 + x)) out.put_line(s, d * 9 + x, b8 + d);
        if (u < 10 && !has(own, u * 9 + x)) out.put_line(s, u * 9 + x, b8 + u - 1);
        if (hit && out.hit >= 0 && *hit < 0) { *hit = out.hit; if (hit_from) *hit_from = out.hit_from; }
        return out.n;
    }
    // stepping pieces: one table-driven loop (:264-286)
    int fly = -1;                                         // flying-king capture target, -1 if none
    if (p == KING) {
        const uint32_t um = file_bits(occ, x) >> (y + 1);
        if (um) {
            const int u = y + 1 + low_bit(um);
            if (has(oking, u * 9 + x)) fly = u * 9 + x;
        }
    }
    const int nd = step_count_imm(p);
    const uint64_t codes = step_row(p);
    const bool ae = EMIT && formula_labels && (p == ADVISOR || p == ELEPHANT);
    const uint64_t ae_labs = ae ? ae_label_row(p == ELEPHANT, s) : 0ull;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int k = 0; k < nd; ++k) {
        const int code = (int)((codes >> (8 * k)) & 0xFFu);
        const int dx = (code & 7) - 2, dy = (code >> 3) - 2;
        const int x_ = x + dx, y_ = y + dy;
        if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9) continue;           // can_move, :323-330
        const int t = y_ * 9 + x_;
        if (has(own, t)) continue;
        if (p == PAWN) {
            if (y < 5 && x_ != x) continue;                           // :270
        } else if (p == KNIGHT || p == ELEPHANT) {
            if (has(occ, (y + dy / 2) * 9 + x + dx / 2)) continue;    // leg / eye; int(d/2) truncates toward 0
            if (p == ELEPHANT && y_ > 4) continue;                    // :275
        } else {                                                      // king, advisor: palace (:277-281)
            if (x_ < 3 || x_ > 5 || y_ > 2) continue;
        }
        if (ae) out.put_labelled(s, t, (uint16_t)((ae_labs >> (16 * k)) & 0xFFFFu));
        else out.put(s, t);
        if (fly >= 0) out.put(s, fly);                                // once per accepted king step (:283-286)
    }
    if (hit && out.hit >= 0 && *hit < 0) { *hit = out.hit; if (hit_from) *hit_from = out.hit_from; }
    return out.n;
}

// ---- a quad of lanes per piece: the wave-level generator (xq_rules.h::wave_movegen) ------------------------------------
// Sub-lane q of a piece's quad does a quarter of its work: ray q of a rook / cannon (0 left, 1 right, 2 down, 3 up), steps
// q and q + 4 of a stepping piece.  Every lane ends up with two segments of the piece's move list: A = the quiet moves of
// its ray / its step q, B = the capture of its ray / its step q + 4.  The reference's order inside a piece is all A segments
// in sub-lane order, then all B segments (quiet l, r, d, u then captures l, r, d, u; steps 0..3 then 4..7), so segment
// offsets are prefix sums over the quad.  quad_plan analyses, quad_emit writes; tests/lane_harness.cpp::lane_movegen_quad
// emulates the cross-lane sums and checks the result against gen_piece.
struct QuadPlan {
    int n_a, n_b;       // moves in segment A / B
    uint32_t st;        // slider: (first blocker coordinate + 1) | (capture coordinate + 1) << 8; stepper: 0x80 | fly square
};

XQ_HD bool step_valid(int p, int x, int y, int code, const Set90& occ, const Set90& own)
{
    const int dx = (code & 7) - 2, dy = (code >> 3) - 2;
    const int x_ = x + dx, y_ = y + dy;
    if (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9) return false;               // can_move, :323-330
    if (has(own, y_ * 9 + x_)) return false;
    if (p == PAWN) return !(y < 5 && x_ != x);                            // :270
    if (p == KNIGHT || p == ELEPHANT) {
        if (has(occ, (y + dy / 2) * 9 + x + dx / 2)) return false;        // leg / eye
        return !(p == ELEPHANT && y_ > 4);                                // :275
    }
    return !(x_ < 3 || x_ > 5 || y_ > 2);                                 // king, advisor: palace (:277-281)
}

XQ_HD QuadPlan quad_plan(int p, int s, int q, const Set90& occ, const Set90& own, const Set90& oking)
{
    const int x = s % 9, y = s / 9;
    QuadPlan pl{0, 0, 0u};
    if (p == ROOK || p == CANNON) {
        const bool horiz = q < 2;
        const uint32_t line = horiz ? rank_bits(occ, y) : file_bits(occ, x);
        const int pos = horiz ? x : y, lim = horiz ? 9 : 10;
        int first, second;                               // along the line; none = -1 towards 0, lim towards the far edge
        if ((q & 1) == 0) {
            uint32_t m = line & ((1u << pos) - 1u);
            first = m ? top_bit(m) : -1;
            if (m) m &= ~(1u << first);
            second = m ? top_bit(m) : -1;                // the next blocker beyond the screen (none without a screen)
            pl.n_a = pos - first - 1;
        } else {
            const uint32_t m = line >> (pos + 1);
            first = m ? pos + 1 + low_bit(m) : lim;
            const uint32_t m2 = m ? (m & (m - 1u)) : 0u;
            second = m2 ? pos + 1 + low_bit(m2) : lim;
            pl.n_a = first - pos - 1;
        }
        const int c = (p == CANNON) ? second : first;
        const bool on = (q & 1) == 0 ? c > -1 : c < lim;
        const int csq = horiz ? y * 9 + c : c * 9 + x;
        pl.n_b = (on && !has(own, csq)) ? 1 : 0;
        pl.st = (uint32_t)(first + 1) | ((uint32_t)(c + 1) << 8);
        return pl;
    }
    int fly = -1;
    if (p == KING) {
        const uint32_t um = file_bits(occ, x) >> (y + 1);
        if (um) {
            const int u = y + 1 + low_bit(um);
            if (has(oking, u * 9 + x)) fly = u * 9 + x;
        }
    }
    const int nd = step_count_imm(p);
    const uint64_t codes = step_row(p);
    const int mult = fly >= 0 ? 2 : 1;                   // the fly move follows every accepted king step
    if (q < nd && step_valid(p, x, y, (int)((codes >> (8 * q)) & 0xFFu), occ, own)) pl.n_a = mult;
    if (q + 4 < nd && step_valid(p, x, y, (int)((codes >> (8 * (q + 4))) & 0xFFu), occ, own)) pl.n_b = mult;
    pl.st = fly >= 0 ? (0x80u | (uint32_t)fly) : 0u;
    return pl;
}

// off_a / off_b: list positions of this lane's two segments
XQ_HD void quad_emit(int p, int s, int q, const QuadPlan& pl, uint16_t* lab, uint16_t* ft, int off_a, int off_b,
                     bool formula_labels, int cap = MAXMOVES)
{
    const int x = s % 9, y = s / 9;
    MoveSink<true> out{lab, ft, off_a, 0, cap, -1, -1, -1, formula_labels && p != ADVISOR && p != ELEPHANT, 0, 0u};
    if (out.formula) label_block(s, &out.base, &out.kvalid);
    if (p == ROOK || p == CANNON) {
        const bool horiz = q < 2;
        const int pos = horiz ? x : y;
        const int first = (int)(pl.st & 0xFFu) - 1, c = (int)((pl.st >> 8) & 0xFFu) - 1;
        const int lo = (q & 1) == 0 ? first + 1 : pos + 1, hi = (q & 1) == 0 ? pos : first;     // quiet coordinates [lo, hi)
        const int lb = horiz ? out.base : out.base + 8;                                         // label of coordinate t: lb + (t < pos ? t : t - 1)
        for (int t = lo; t < hi; ++t) out.put_line(s, horiz ? y * 9 + t : t * 9 + x, lb + (t < pos ? t : t - 1));
        if (pl.n_b) {
            out.off = off_b; out.n = 0;
            out.put_line(s, horiz ? y * 9 + c : c * 9 + x, lb + (c < pos ? c : c - 1));
        }
        return;
    }
    const int fly = (pl.st & 0x80u) ? (int)(pl.st & 0x7Fu) : -1;
    const uint64_t codes = step_row(p);
    const bool ae = formula_labels && (p == ADVISOR || p == ELEPHANT);
    const uint64_t ae_labs = ae ? ae_label_row(p == ELEPHANT, s) : 0ull;
    for (int half = 0; half < 2; ++half) {
        if ((half ? pl.n_b : pl.n_a) == 0) continue;
        const int k = q + 4 * half;
        const int code = (int)((codes >> (8 * k)) & 0xFFu);
        const int t = (y + (code >> 3) - 2) * 9 + x + (code & 7) - 2;
        out.off = half ? off_b : off_a; out.n = 0;
        if (ae) out.put_labelled(s, t, (uint16_t)((ae_labs >> (16 * k)) & 0xFFFFu));
        else out.put(s, t);
        if (fly >= 0) out.put(s, fly);
    }
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
