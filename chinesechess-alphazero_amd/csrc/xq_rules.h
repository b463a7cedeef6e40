// xq_rules.h -- wave-cooperative Xiangqi rules for gfx950 (device only).
//
// Execution model: ONE 64-lane wavefront per board (workgroup = 64 threads), the board and the ordered move
// lists staged in that wave's LDS.  The board is reduced to three 90-bit square sets by ballots and lane r
// generates the moves of the mover's r-th piece (xq_lane.h::gen_piece); the reference's move ORDER
// (static_env.py:256-321: squares y-major/x-minor, per-piece direction order) is kept by a two-pass ordered
// compaction: count per piece -> wave prefix sum -> emit at the piece's offset.  Terminal detection, check
// detection and the perpetual check/chase helpers are ballots over those lists.
//
// Synchronisation inside the single-wave workgroup: see wave_sync() / wave_sync_global() below.
#pragma once
#include <hip/hip_runtime.h>
#include "xq_lane.h"

namespace xq {

constexpr int BOARD_LDS = 96;   // 90 squares padded to a multiple of 16 B

struct MoveList {               // ordered move list in LDS
    uint16_t lab[MAXMOVES];     // label index
    uint16_t ft[MAXMOVES];      // from << 8 | to
};

// Per-wave LDS scratch for the rules (about 2.6 KB).
struct RulesLDS {
    int8_t bd[4][BOARD_LDS];    // 0: position, 1..3: derived boards (flip / step / nested step)
    MoveList ml[3];
    uint32_t cset[2][MAXMOVES]; // chase sets of will_check_or_catch
    uint16_t plist[BOARD_LDS];  // the mover's pieces in square order: square | type << 8
};

XQ_D int lane_id() { return (int)(threadIdx.x & 63u); }
// Two kinds of intra-wave synchronisation (a workgroup is ONE wavefront):
//  * wave_sync(): LDS traffic only.  The LDS unit executes a wave's DS instructions in order, so lanes see each
//    other's LDS writes as soon as the compiler keeps the program order: a compiler barrier + lgkmcnt(0).
//    Outstanding GLOBAL loads / stores are NOT waited for (a full __syncthreads() emits `s_waitcnt vmcnt(0)`,
//    a complete HBM round trip, at every call).
//  * wave_sync_global(): also orders global memory between lanes (workgroup-scope fence = vmcnt(0)).
//    Needed only where a lane reads global data another lane of the wave wrote in the same launch.
XQ_D void wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
XQ_D void wave_sync_global() { __syncthreads(); }

// inclusive prefix sum over the wave's 64 lanes: Kogge-Stone inside each row of 16 lanes with DPP row shifts (a lane
// without a source adds 0), then the row totals carried across with row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2,
// 3): six VALU adds, no LDS permute.  (Round 2 ran six __shfl_up steps -- ds_bpermute round trips; this and the quad
// generator below took the sustained search round from 0.388 to 0.343 ms in the A/B of round 3, all GPU suites bit-exact.)
XQ_D int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2 and 3
    return v;
}

// Wave-wide reductions on the same DPP ladder (round 5): the running value of lane 63 after the six steps is the
// reduction over all 64 lanes; one v_readlane hands it to everybody as a scalar.  A lane without a source keeps its own
// value (old = the value itself, bound_ctrl off), which is neutral for max / min.  Call them in wave-uniform control flow (all
// 64 lanes active: the result is read from lane 63 / lane 15).  These replace the six-stage
// __shfl_xor butterflies (two ds_bpermute round trips per stage for a double) on the PUCT arg-max of every tree level.
#define XQ_DPP_STEP_MAX_F64(v, ctrl, rmask)                                                                          \
    do {                                                                                                             \
        const long long b_ = __double_as_longlong(v);                                                               \
        const int lo_ = __builtin_amdgcn_update_dpp((int)b_, (int)b_, ctrl, rmask, 0xF, false);                     \
        const int hi_ = __builtin_amdgcn_update_dpp((int)(b_ >> 32), (int)(b_ >> 32), ctrl, rmask, 0xF, false);     \
        const double o_ = __longlong_as_double((long long)(((unsigned long long)(unsigned int)hi_ << 32) | (unsigned int)lo_)); \
        v = o_ > v ? o_ : v;                                                                                         \
    } while (0)
XQ_D double wave_max_f64(double v)          // v is never NaN
{
    XQ_DPP_STEP_MAX_F64(v, 0x111, 0xF);
    XQ_DPP_STEP_MAX_F64(v, 0x112, 0xF);
    XQ_DPP_STEP_MAX_F64(v, 0x114, 0xF);
    XQ_DPP_STEP_MAX_F64(v, 0x118, 0xF);
    XQ_DPP_STEP_MAX_F64(v, 0x142, 0xA);
    XQ_DPP_STEP_MAX_F64(v, 0x143, 0xC);
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#undef XQ_DPP_STEP_MAX_F64
#define XQ_DPP_STEP_MAX_F32(v, ctrl, rmask)                                                                          \
    do {                                                                                                             \
        const float o_ = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xF, false)); \
        v = o_ > v ? o_ : v;                                                                                         \
    } while (0)
XQ_D float wave_max_f32(float v)            // v is never NaN
{
    XQ_DPP_STEP_MAX_F32(v, 0x111, 0xF);
    XQ_DPP_STEP_MAX_F32(v, 0x112, 0xF);
    XQ_DPP_STEP_MAX_F32(v, 0x114, 0xF);
    XQ_DPP_STEP_MAX_F32(v, 0x118, 0xF);
    XQ_DPP_STEP_MAX_F32(v, 0x142, 0xA);
    XQ_DPP_STEP_MAX_F32(v, 0x143, 0xC);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef XQ_DPP_STEP_MAX_F32
// xor over lanes 0 .. 15 (the first DPP row), returned as a scalar: lane 15 of the row's inclusive xor scan
XQ_D uint32_t row0_xor_u32(uint32_t v)
{
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}

XQ_D int highest_bit(uint64_t lo, uint64_t hi)   // index over the 128-bit (lo: 0..63, hi: 64..127), -1 if none
{
    if (hi) return 64 + 63 - __clzll((long long)hi);
    if (lo) return 63 - __clzll((long long)lo);
    return -1;
}
XQ_D int lowest_bit(uint64_t lo, uint64_t hi)
{
    if (lo) return __ffsll((long long)lo) - 1;
    if (hi) return 64 + __ffsll((long long)hi) - 1;
    return -1;
}

// Load one int8[90] board from global memory into LDS (2-byte loads: 90*i is even).
XQ_D void load_board(const int8_t* __restrict__ g, int8_t* b)
{
    const int lane = lane_id();
    if (lane < 45) {
        const uint16_t w = reinterpret_cast<const uint16_t*>(g)[lane];
        reinterpret_cast<uint16_t*>(b)[lane] = w;
    }
    if (lane >= 45 && lane < 48) reinterpret_cast<uint16_t*>(b)[lane] = 0;
    wave_sync();
}

XQ_D void store_board(const int8_t* b, int8_t* __restrict__ g)
{
    const int lane = lane_id();
    if (lane < 45) reinterpret_cast<uint16_t*>(g)[lane] = reinterpret_cast<const uint16_t*>(b)[lane];
}

// fliped_state (static_env.py:245-254): out[89-s] = -in[s]
XQ_D void flip_board(const int8_t* in, int8_t* out)
{
    const int lane = lane_id();
    out[89 - lane] = (int8_t)(-in[lane]);
    if (lane < 26) out[25 - lane] = (int8_t)(-in[lane + 64]);
    wave_sync();
}

// step (static_env.py:79-86): move the piece, then flip.  Caller checked in[from] != 0.
XQ_D void step_board(const int8_t* in, int from, int to, int8_t* out)
{
    const int lane = lane_id();
    {
        int p = in[lane];
        if (lane == to) p = in[from];
        if (lane == from) p = 0;
        out[89 - lane] = (int8_t)(-p);
    }
    if (lane < 26) {
        const int s = lane + 64;
        int p = in[s];
        if (s == to) p = in[from];
        if (s == from) p = 0;
        out[89 - s] = (int8_t)(-p);
    }
    wave_sync();
}

// The position after `from->to` WITHOUT the perspective flip (== fliped_state(step(.)))
XQ_D void apply_move_noflip(const int8_t* in, int from, int to, int8_t* out)
{
    const int lane = lane_id();
    {
        int p = in[lane];
        if (lane == to) p = in[from];
        if (lane == from) p = 0;
        out[lane] = (int8_t)p;
    }
    if (lane < 26) {
        const int s = lane + 64;
        int p = in[s];
        if (s == to) p = in[from];
        if (s == from) p = 0;
        out[s] = (int8_t)p;
    }
    wave_sync();
}

// get_legal_moves (static_env.py:256-321).  Returns the move count (may exceed MAXMOVES only for impossible
// boards; entries beyond MAXMOVES are dropped).  Board must be visible in LDS.
// The board is reduced to three wave-uniform square sets by ballots; the mover's pieces are compacted (in
// square order) so that lane r generates the moves of the r-th piece from those sets alone, and the ordered
// move list is assembled by a prefix sum over the per-piece counts.
// FORMULA: labels by arithmetic (label_of_line_or_knight) instead of the table gather -- for latency-bound callers.
template <bool FORMULA = false>
XQ_D int wave_movegen(const int8_t* b, MoveList& ml, uint16_t* plist)
{
    const int lane = lane_id();
    const int p0 = b[lane];
    const int p1 = (lane < 26) ? b[lane + 64] : 0;
    const Set90 occ{__ballot(p0 != 0), __ballot(p1 != 0)};
    const Set90 own{__ballot(p0 > 0), __ballot(p1 > 0)};
    const Set90 oking{__ballot(p0 == -KING), __ballot(p1 == -KING)};
    const uint64_t below = (1ull << lane) - 1ull;
    const int n_lo = __popcll(own.lo), np = n_lo + __popcll(own.hi);
    wave_sync();                                  // earlier readers of `ml` / `plist` are done
    if (p0 > 0) plist[__popcll(own.lo & below)] = (uint16_t)(lane | (p0 << 8));
    if (p1 > 0) plist[n_lo + __popcll(own.hi & below)] = (uint16_t)((lane + 64) | (p1 << 8));
    wave_sync();
    int total = 0;
    // a quad of lanes per piece, 16 pieces per pass (every legal position: one pass): a slider's four rays / a stepper's
    // steps k and k + 4 on the four lanes of the quad (quad_plan / quad_emit in xq_lane.h), so all 64 lanes work where one
    // lane per piece kept 16 busy
    for (int base = 0; base < np; base += 16) {
        const int r = lane >> 2, q = lane & 3;
        const bool act = base + r < np;
        const int e = act ? plist[base + r] : 0;
        const int s = e & 0xFF, p = e >> 8;
        const QuadPlan pl = act ? quad_plan(p, s, q, occ, own, oking) : QuadPlan{0, 0, 0u};
        // the quad's segment sizes on every lane of the quad: DPP quad_perm broadcasts (one VALU move each)
        const int mine = pl.n_a | (pl.n_b << 8);
        int off_a = 0, off_b = 0, sum_a = 0, sum_b = 0;
        const int v4[4] = {__builtin_amdgcn_mov_dpp(mine, 0x00, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(mine, 0x55, 0xF, 0xF, true),
                           __builtin_amdgcn_mov_dpp(mine, 0xAA, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(mine, 0xFF, 0xF, 0xF, true)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = v4[k];
            if (k < q) { off_a += v & 0xFF; off_b += v >> 8; }
            sum_a += v & 0xFF; sum_b += v >> 8;
        }
        const int n = sum_a + sum_b;
        const int inc = wave_incl_scan(q == 0 ? n : 0);            // at lane 4 r + q: the pieces up to and including r
        const int poff = total + inc - n;
        if (mine) quad_emit(p, s, q, pl, ml.lab, ml.ft, poff + off_a, poff + sum_a + off_b, FORMULA);
        total += __builtin_amdgcn_readlane(inc, 63);
    }
    wave_sync();
    return total;
}

// index of the first move in ml[0..n) whose destination is `sq`, or -1
XQ_D int first_move_to(const MoveList& ml, int n, int sq)
{
    const int lane = lane_id();
    n = n < MAXMOVES ? n : MAXMOVES;
    const uint64_t m0 = __ballot(lane < n && (ml.ft[lane] & 0xFF) == sq);
    const uint64_t m1 = __ballot(lane + 64 < n && (ml.ft[lane + 64] & 0xFF) == sq);
    return lowest_bit(m0, m1);
}

struct DoneResult {
    int over;        // 0/1
    int v;           // value for the side to move
    int final_move;  // label of the first king-capturing move, NOMOVE when None
    int check;       // only with need_check
    int nmoves;      // >= 0: ml0 holds the position's ordered move list; -1: not generated
};

// done (static_env.py:14-77).  b: position; tmpb: scratch board; ml0: receives the move list of b
// (when the early tests did not decide); ml1: scratch list for the need_check pass.
template <bool FORMULA = false>
XQ_D DoneResult wave_done(const int8_t* b, int8_t* tmpb, MoveList& ml0, MoveList& ml1, uint16_t* plist, bool need_check)
{
    const int lane = lane_id();
    DoneResult r{0, 0, NOMOVE, 0, -1};
    const int p0 = b[lane];
    const int p1 = (lane < 26) ? b[lane + 64] : 0;
    const uint64_t ok0 = __ballot(p0 == -KING), ok1 = __ballot(p1 == -KING);
    const uint64_t mk0 = __ballot(p0 == KING), mk1 = __ballot(p1 == KING);
    if (!(ok0 | ok1)) { r.over = 1; r.v = 1; return r; }       // 's' not in state
    if (!(mk0 | mk1)) { r.over = 1; r.v = -1; return r; }      // 'S' not in state
    const int rk = highest_bit(mk0, mk1), bk = highest_bit(ok0, ok1);   // scan order: last one wins
    const int rx = rk % 9, ry = rk / 9, bx = bk % 9, by = bk / 9;
    int winner = 0;
    if (ry == 0 && rx == 0) { winner = 2; r.v = -1; }          // dead branches kept, :33-38
    else if (by == 0 && bx == 0) { winner = 1; r.v = 1; }
    else if (rx == bx) {                                       // kings on one file, :39-49
        const int x0 = lane % 9, y0 = lane / 9;
        const int s1 = lane + 64, x1 = s1 % 9, y1 = s1 / 9;
        const bool blk0 = (x0 == rx && y0 > ry && y0 < by && p0 != 0);
        const bool blk1 = (lane < 26 && x1 == rx && y1 > ry && y1 < by && p1 != 0);
        if (!__ballot(blk0 || blk1)) { r.v = 1; winner = 1; }
    }
    if (!winner) {                                             // :52-60
        const int n = wave_movegen<FORMULA>(b, ml0, plist);
        r.nmoves = n;
        const int k = first_move_to(ml0, n, bk);
        if (k >= 0) { winner = 1; r.v = 1; r.final_move = ml0.lab[k]; }
    }
    if (!winner && need_check) {                               // :61-73
        flip_board(b, tmpb);
        const int n2 = wave_movegen<FORMULA>(tmpb, ml1, plist);
        r.check = first_move_to(ml1, n2, 89 - rk) >= 0;
    }
    r.over = winner != 0;
    return r;
}

// has_attack_chessman (static_env.py:471-479): any rook / knight / pawn / cannon of either side
XQ_D int wave_has_attack(const int8_t* b)
{
    const int lane = lane_id();
    int t0 = b[lane]; t0 = t0 < 0 ? -t0 : t0;
    int t1 = (lane < 26) ? b[lane + 64] : 0; t1 = t1 < 0 ? -t1 : t1;
    const bool a0 = (t0 == ROOK || t0 == KNIGHT || t0 == PAWN || t0 == CANNON);
    const bool a1 = (t1 == ROOK || t1 == KNIGHT || t1 == PAWN || t1 == CANNON);
    return __ballot(a0 || a1) != 0;
}

// be_catched (static_env.py:456-469): is the square the move starts from attacked right now
XQ_D int wave_be_catched(const int8_t* b, int from, int8_t* tmpb, MoveList& ml, uint16_t* plist)
{
    flip_board(b, tmpb);
    const int n = wave_movegen<true>(tmpb, ml, plist);        // only the squares of the list are read: cheap labels
    return first_move_to(ml, n, 89 - from) >= 0;
}

// get_catch_list (static_env.py:423-454) over a given ordered move list.  Keys:
// attacker type << 24 | from << 16 | victim type << 8 | to.  Returns the set size.
XQ_D int wave_catch_list(const int8_t* b, const MoveList& moves, int nmoves,
                         int8_t* nextb, MoveList& reply, uint32_t* set, uint16_t* plist)
{
    const int lane = lane_id();
    int cnt = 0;
    nmoves = nmoves < MAXMOVES ? nmoves : MAXMOVES;
    for (int k = 0; k < nmoves; ++k) {
        const int ft = moves.ft[k];
        const int f = ft >> 8, t = ft & 0xFF;
        const int vict = b[t];
        if (vict == 0) continue;                               // no capture
        // (the three cheap exclusions first: the reference tests them after be_catched, all four only skip the move)
        const int a = b[f];
        if (a == PAWN && f / 9 <= 4) continue;                 // :443-444
        if (vict == -PAWN && t / 9 > 4) continue;              // :447-448
        if (-vict == a) continue;                              // exchange, :450-451
        step_board(b, f, t, nextb);
        const int nr = wave_movegen<true>(nextb, reply, plist);
        if (first_move_to(reply, nr, 89 - t) >= 0) continue;   // could be recaptured
        const uint32_t key = ((uint32_t)a << 24) | ((uint32_t)f << 16) | ((uint32_t)(-vict) << 8) | (uint32_t)t;
        const bool dup0 = lane < cnt && set[lane] == key;
        const bool dup1 = lane + 64 < cnt && set[lane + 64] == key;
        if (!__ballot(dup0 || dup1)) {
            if (lane == 0) set[cnt] = key;
            ++cnt;
        }
        wave_sync();
    }
    return cnt;
}

// will_check_or_catch (static_env.py:390-421).  b is RulesLDS::bd[0]; returns -1 when the move's
// source square is empty (the reference raises ValueError).
XQ_D int wave_will_check_or_catch(RulesLDS& w, const int8_t* b, int label)
{
    const int lane = lane_id();
    const int ft = label_ft(label);
    const int f = ft >> 8, t = ft & 0xFF;
    if (b[f] == 0) return -1;
    int8_t* black = w.bd[1];
    apply_move_noflip(b, f, t, black);            // == fliped_state(step(ori_state, action))
    // the opponent's king as `state` sees it: last 'k' in scan order of `state` == lowest square here
    const int q0 = black[lane], q1 = (lane < 26) ? black[lane + 64] : 0;
    const int ksq = lowest_bit(__ballot(q0 == -KING), __ballot(q1 == -KING));
    const int target = ksq >= 0 ? ksq : 89;       // red_k stays [0,0] -> (9,8) when the king is gone
    const int nb = wave_movegen<true>(black, w.ml[0], w.plist);
    if (first_move_to(w.ml[0], nb, target) >= 0) return 1;                        // :406-411
    const int n1 = wave_movegen<true>(b, w.ml[1], w.plist);
    const int c1 = wave_catch_list(b, w.ml[1], n1, w.bd[2], w.ml[2], w.cset[0], w.plist);  // first_set
    const int c2 = wave_catch_list(black, w.ml[0], nb, w.bd[2], w.ml[2], w.cset[1], w.plist);
    // second_set - first_set != {} and len(second_set) >= len(first_set), :415
    bool fresh = false;
    for (int i = 0; i < c2; ++i) {
        const uint32_t key = w.cset[1][i];
        const bool in0 = lane < c1 && w.cset[0][lane] == key;
        const bool in1 = lane + 64 < c1 && w.cset[0][lane + 64] == key;
        if (!__ballot(in0 || in1)) { fresh = true; break; }
    }
    return (fresh && c2 >= c1) ? 1 : 0;
}

// state_to_planes (static_env.py:137-156) for one board, written as 315 chunks of 4 elements.
// DT: 0 = f32, 1 = f16, 2 = bf16, 3 = u8.  0/1 are exact in every format.
template <int DT>
XQ_D void wave_encode(const int8_t* b, void* __restrict__ out)
{
    const int lane = lane_id();
    for (int q = lane; q < 315; q += 64) {
        const int o = q * 4;
        const int b0 = plane_bit(b, o), b1 = plane_bit(b, o + 1), b2 = plane_bit(b, o + 2), b3 = plane_bit(b, o + 3);
        if (DT == 0) {
            float4 v = make_float4((float)b0, (float)b1, (float)b2, (float)b3);
            reinterpret_cast<float4*>(out)[q] = v;
        } else if (DT == 1 || DT == 2) {
            const uint32_t one = (DT == 1) ? 0x3C00u : 0x3F80u;    // 1.0 in f16 / bf16
            uint2 v;
            v.x = (b0 ? one : 0u) | ((b1 ? one : 0u) << 16);
            v.y = (b2 ? one : 0u) | ((b3 ? one : 0u) << 16);
            reinterpret_cast<uint2*>(out)[q] = v;
        } else {
            reinterpret_cast<uint32_t*>(out)[q] = (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16) | ((uint32_t)b3 << 24);
        }
    }
}

// Same planes through a per-board code row: codes[pos] = channel (0..13) of the piece on the square that plane
// position pos = i*9 + j shows (row i of the planes is y = 9 - i), 0xFF for an empty square.  One LDS byte write
// per square, then each output element is a byte compare instead of two divisions and a board lookup.
template <int DT>
XQ_D void wave_encode_codes(const int8_t* b, uint8_t* codes, void* __restrict__ out)
{
    const int lane = lane_id();
    wave_sync();
    for (int s = lane; s < NSQ; s += 64) {
        const int p = b[s];
        const int y = s / 9, x = s - y * 9;
        codes[(9 - y) * 9 + x] = (uint8_t)(p == 0 ? 0xFF : (p > 0 ? p - 1 : 6 - p));
    }
    wave_sync();
    for (int q = lane; q < 315; q += 64) {
        const int o = q * 4;
        int c = o / 90, pos = o - c * 90;
        uint32_t bit[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bit[e] = codes[pos] == c;
            if (++pos == 90) { pos = 0; ++c; }
        }
        if (DT == 0) {
            reinterpret_cast<float4*>(out)[q] = make_float4((float)bit[0], (float)bit[1], (float)bit[2], (float)bit[3]);
        } else if (DT == 1 || DT == 2) {
            const uint32_t one = (DT == 1) ? 0x3C00u : 0x3F80u;
            uint2 v;
            v.x = (bit[0] ? one : 0u) | ((bit[1] ? one : 0u) << 16);
            v.y = (bit[2] ? one : 0u) | ((bit[3] ? one : 0u) << 16);
            reinterpret_cast<uint2*>(out)[q] = v;
        } else {
            reinterpret_cast<uint32_t*>(out)[q] = bit[0] | (bit[1] << 8) | (bit[2] << 16) | (bit[3] << 24);
        }
    }
}

}  // namespace xq
