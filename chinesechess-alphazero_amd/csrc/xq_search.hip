// xq_search.hip -- batched PUCT-MCTS + self-play game loop for gfx950, one wavefront per game.
//
// Replaces, for thousands of concurrent games, the reference's per-game Python objects:
//   CChessPlayer.action / MCTS_search / select_action_q_and_u / expand_and_evaluate / update_tree /
//   calc_policy / apply_temperature      (cchess_alphazero/agent/player.py:145-470)
//   SelfPlayWorker.start_game            (cchess_alphazero/worker/self_play.py:95-212)
//
// cz_search_round() is one lock-step ROUND for every game, three launches on one stream:
//   k_sim(BACKUP)  1. attach the network results of the previous round to their leaves and back them up,
//                  2. resume simulations that were parked on those leaves;
//   k_advance      3. where a search is complete: pick the move, apply the game rules, start the next
//                     search / the next game (self-play) or mark the game READY (external mode);
//   k_sim(SELECT)  4. where a batch of K simulations is complete start the next one; every new leaf writes
//                     its input planes into its fixed slot (game * K + sim) of the evaluation queue.
// The host then runs ONE network forward over the whole queue.
// There is no host decision inside a round and no device->host copy, so a round (kernel + network
// forward) can be replayed from a HIP graph.
//
// Arithmetic follows the reference bit for bit: priors float32 (summed in move order),
// sqrt / Q / U in float64 with the float32 product c_puct*p for non-root nodes, W float64.
// Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "xq_rules.h"
#include "xq_search.h"
#include "xq_noise.h"
#include "../../include/czero.h"

using namespace xq;

extern "C" void czi_set_error(const char* msg);   // xq_kernels.hip

namespace {

constexpr int MAXD_LDS = 128;          // LDS copy of the current path; SearchParams.max_depth <= this
constexpr int INIT_NIB_WORDS = KEY_WORDS;


struct SearchLDS {
    RulesLDS r;
    uint32_t key[KEY_WORDS + 4];
    int32_t path_node[MAXD_LDS];
    int32_t path_edge[MAXD_LDS];
    unsigned long long ctr[CT_COUNT];  // counter accumulators of the running kernel
    uint8_t codes[BOARD_LDS];          // plane codes of the leaf being encoded
    double dsc[MAXMOVES];              // sampling scratch
    double cdf[MAXMOVES];
    float pr[MAXMOVES];                // prior gather scratch
    uint16_t slab[MAXMOVES];           // labels sorted
    int32_t sn[MAXMOVES];              // visit counts sorted by label
    uint32_t chtab[MAX_CHUNKS];        // the game's chunk table (pool chunk number of local chunk i)
};

// one edge's visit statistics (ActionState n / w + the child link); edge j of a node = granule stat + j
struct __attribute__((aligned(16))) EdgeStat {
    double w;
    int32_t n;
    int32_t child;          // node id; CHILD_UNKNOWN / CHILD_TERM_*
};
static_assert(sizeof(EdgeStat) == 16, "one granule per edge");

// node record: key[12] | {sum_n, meta, stat, -} | float p[nm] | uint16 mv[nm]
constexpr int NODE_OFF_HDR = 48, NODE_OFF_P = 64;

// pointers into one game's state
struct GameView {
    char* pool;
    const uint32_t* chtab;  // LDS copy of the chunk table
    uint64_t* hash;
    int32_t* path_node;     // [K][max_depth]
    int32_t* path_edge;
    uint8_t* s_state;
    int32_t* s_depth;
    int32_t* s_node;
    unsigned long long* ctr;     // global per-game counters
    unsigned long long* lctr;    // this kernel's LDS accumulators
    int g;
};

XQ_D int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
XQ_D uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
XQ_D uint64_t uni64(uint64_t v)
{
    const uint32_t lo = uniu((uint32_t)v), hi = uniu((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
XQ_D double unid(double v) { return __longlong_as_double((long long)uni64((uint64_t)__double_as_longlong(v))); }

// `chtab` = LDS array of MAX_CHUNKS words that receives the game's chunk table
XQ_D GameView make_view(const SearchBuffers& B, const SearchParams& P, int g, unsigned long long* lctr, uint32_t* chtab)
{
    GameView v;
    v.lctr = lctr;
    v.pool = B.pool;
    const int nch = uni(B.g_nchunks[g]);
    for (int i = lane_id(); i < nch; i += 64) chtab[i] = B.g_chunk_tab[(size_t)g * P.max_chunks + i];
    v.chtab = chtab;
    v.hash = B.hash_tab + (size_t)g * P.hash_cap;
    v.path_node = B.s_path_node + (size_t)g * P.K * P.max_depth;
    v.path_edge = B.s_path_edge + (size_t)g * P.K * P.max_depth;
    v.s_state = B.s_state + (size_t)g * P.K;
    v.s_depth = B.s_depth + (size_t)g * P.K;
    v.s_node = B.s_node + (size_t)g * P.K;
    v.ctr = B.counters + (size_t)g * CT_COUNT;
    v.g = g;
    wave_sync();
    return v;
}

// record id -> address (ids of one record never straddle a chunk)
XQ_D char* rec_ptr(const GameView& gv, uint32_t id)
{
    return gv.pool + ((size_t)gv.chtab[id >> CHUNK_SHIFT] << 20) + ((size_t)(id & (uint32_t)(CHUNK_GRANULES - 1)) << 4);
}
XQ_D const uint32_t* node_key(const GameView& gv, int node) { return reinterpret_cast<const uint32_t*>(rec_ptr(gv, (uint32_t)node)); }
XQ_D float* node_p(char* base) { return reinterpret_cast<float*>(base + NODE_OFF_P); }
XQ_D uint16_t* node_mv(char* base, int nm) { return reinterpret_cast<uint16_t*>(base + NODE_OFF_P + 4 * nm); }
XQ_D EdgeStat* edge_ptr(const GameView& gv, uint32_t edge) { return reinterpret_cast<EdgeStat*>(rec_ptr(gv, edge)); }

// Counters accumulate in LDS while a kernel runs (no global round trip per event) and are flushed once.
XQ_D void count(const GameView& gv, int which, unsigned long long by = 1)
{
    if (lane_id() == 0) gv.lctr[which] += by;
}
XQ_D void count_max(const GameView& gv, int which, unsigned long long val)
{
    if (lane_id() == 0 && gv.lctr[which] < val) gv.lctr[which] = val;
}
#ifdef CZ_SIM_PROFILE
#define PROF_BEGIN() long long prof_t = clock64()
#define PROF(which) do { const long long t_ = clock64(); count(gv, which, (unsigned long long)(t_ - prof_t)); prof_t = t_; } while (0)
#else
#define PROF_BEGIN() do { } while (0)
#define PROF(which) do { } while (0)
#endif
XQ_D void counters_begin(const GameView& gv)
{
    for (int i = lane_id(); i < CT_COUNT; i += 64) gv.lctr[i] = 0;
    wave_sync();
}
XQ_D void counters_flush(const GameView& gv)
{
    wave_sync_global();
    for (int i = lane_id(); i < CT_COUNT; i += 64) {
        const unsigned long long v = gv.lctr[i];
        if (v == 0) continue;
        if (i == CT_MAX_DEPTH) { if (gv.ctr[i] < v) gv.ctr[i] = v; }
        else gv.ctr[i] += v;
    }
}

// ---- counter-based RNG (Philox4x32-10), same stream as oracle/xq_mcts.c ------------------------
XQ_D double philox_uniform(uint64_t seed, uint32_t game_id, uint32_t stream, uint64_t idx)
{
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream, c3 = game_id;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6)) / 9007199254740992.0;
}

// (the root noise generator -- NoiseRng, gamma_draw, dirichlet0 -- lives in xq_noise.h: host + device, CPU-tested)

// ---- packed keys and the transposition hash ----------------------------------------------------
// board (LDS) -> key words in L.key (lanes 0..11), returns the 64-bit hash (wave-uniform)
XQ_D uint64_t pack_key(const int8_t* b, uint32_t* key)
{
    const int lane = lane_id();
    uint64_t h = 0;
    if (lane < KEY_WORDS) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = lane * 8 + j;
            const int p = s < NSQ ? b[s] : 0;
            w |= nib_of(p) << (4 * j);
        }
        key[lane] = w;
        h = mix64((uint64_t)w + 0x9E3779B97F4A7C15ULL * (uint64_t)(lane + 1));
    }
    // xor of the twelve word hashes (lanes 12 .. 15 hold 0): a DPP scan of the first row, no LDS permute
    const uint32_t lo = row0_xor_u32((uint32_t)h), hi = row0_xor_u32((uint32_t)(h >> 32));
    wave_sync();
    h = ((uint64_t)hi << 32) | lo;
    return mix64(h);
}

// hash of an already packed key (same value pack_key returns for that position)
XQ_D uint64_t hash_of_key(const uint32_t* key)
{
    const int lane = lane_id();
    uint64_t h = 0;
    if (lane < KEY_WORDS) h = mix64((uint64_t)key[lane] + 0x9E3779B97F4A7C15ULL * (uint64_t)(lane + 1));
    const uint32_t lo = row0_xor_u32((uint32_t)h), hi = row0_xor_u32((uint32_t)(h >> 32));
    h = ((uint64_t)hi << 32) | lo;
    return mix64(h);
}

XQ_D void unpack_key(const uint32_t* __restrict__ gkey, int8_t* b)
{
    const int lane = lane_id();
    {
        const uint32_t w = gkey[lane >> 3];
        b[lane] = (int8_t)piece_of_nib((w >> (4 * (lane & 7))) & 15u);
    }
    if (lane < 26) {
        const int s = lane + 64;
        const uint32_t w = gkey[s >> 3];
        b[s] = (int8_t)piece_of_nib((w >> (4 * (s & 7))) & 15u);
    }
    if (lane >= 26 && lane < 32) b[lane + 64] = 0;
    wave_sync();
}

// returns node index or -1; *slot_out = where to insert
XQ_D int hash_lookup(const GameView& gv, const SearchParams& P, const uint32_t* key, uint64_t h, int* slot_out)
{
    const int lane = lane_id();
    const uint32_t mask = (uint32_t)P.hash_cap - 1u;
    uint32_t slot = (uint32_t)h & mask;
    const uint32_t tag = (uint32_t)(h >> 32) | 1u;
    for (int probe = 0; probe < P.hash_cap; ++probe) {
        const uint64_t e = uni64(gv.hash[slot]);
        if (e == 0) { *slot_out = (int)slot; return -1; }
        if ((uint32_t)(e >> 32) == tag) {
            const int idx = (int)((uint32_t)e) - 1;
            const bool ne = lane < KEY_WORDS && node_key(gv, idx)[lane] != key[lane];
            if (!__ballot(ne)) { *slot_out = (int)slot; return idx; }
        }
        slot = (slot + 1) & mask;
    }
    *slot_out = -1;
    return -1;
}

// ---- backup: update_tree, player.py:357-366 ------------------------------------------------------
// levels are independent (a path never holds the same edge twice), so lane i updates level i
XQ_D void backup(const SearchParams& P, const GameView& gv, const SearchLDS& L, int depth, double v)
{
    const int lane = lane_id();
    for (int i = lane; i < depth; i += 64) {
        EdgeStat* e = edge_ptr(gv, (uint32_t)L.path_edge[i]);
        const double vi = ((depth - i) & 1) ? -v : v;        // v = -v once per level walking up
        const EdgeStat cur = *e;                             // one 16-byte load
        e->n = cur.n + (1 - P.vl);
        e->w = cur.w + (vi + (double)P.vl);
    }
    count(gv, CT_SIMS);
    count(gv, CT_SUM_DEPTH, (unsigned long long)depth);
    count_max(gv, CT_MAX_DEPTH, (unsigned long long)depth);
    wave_sync_global();
}

// sum_n / (move count | flags) / stat block of a node: one 16-byte load
struct NodeHdr {
    int sum_n;
    uint32_t meta, stat;
};
XQ_D NodeHdr load_hdr(char* base)
{
    // lane 0 is also the only lane that ever writes these words, so its load sees its own earlier stores
    // without a fence
    int4 v = make_int4(0, 0, 0, 0);
    if (lane_id() == 0) v = *reinterpret_cast<const int4*>(base + NODE_OFF_HDR);
    NodeHdr h;
    h.sum_n = __builtin_amdgcn_readfirstlane(v.x);
    h.meta = (uint32_t)__builtin_amdgcn_readfirstlane(v.y);
    h.stat = (uint32_t)__builtin_amdgcn_readfirstlane(v.z);
    return h;
}
XQ_D void store_sum_n(char* base, int v) { *reinterpret_cast<int32_t*>(base + NODE_OFF_HDR) = v; }
XQ_D void store_meta(char* base, uint32_t v) { *reinterpret_cast<uint32_t*>(base + NODE_OFF_HDR + 4) = v; }
XQ_D void store_stat(char* base, uint32_t v) { *reinterpret_cast<uint32_t*>(base + NODE_OFF_HDR + 8) = v; }

// ---- prior spreading: select_action_q_and_u, player.py:272-284 ----------------------------------
// Network rows as raw logits (cz_search_policy_logits): the reference spreads the softmax output over the legal moves,
// p_j / sum_legal p (player.py:272-283), in which the softmax's own denominator cancels -- so only the legal moves' logits
// matter: p_j := exp(l_j - max over the node's moves).  (The engine's own queue then skips the normalising pass over all
// 2086 columns: 546 MB of traffic per round for the 4 % of the entries that are read.)
XQ_D void logits_to_weights(float& q0, float& q1, int nm)
{
    const int lane = lane_id();
    const bool h0 = lane < nm, h1 = lane + 64 < nm;
    float m = h0 ? q0 : -3.0e38f;
    if (h1 && q1 > m) m = q1;
    m = wave_max_f32(m);
    q0 = h0 ? __expf(q0 - m) : 0.0f;
    q1 = h1 ? __expf(q1 - m) : 0.0f;
}

XQ_D void attach_policy(const GameView& gv, SearchLDS& L, int node, const float* __restrict__ prow, bool logits)
{
    const int lane = lane_id();
    char* base = rec_ptr(gv, (uint32_t)node);
    const NodeHdr hdr = load_hdr(base);
    const uint32_t meta = hdr.meta;
    const int nm = (int)(meta & 0xFF);
    float* pp = node_p(base);
    const uint16_t* pm = node_mv(base, nm);
    float q0 = 0.0f, q1 = 0.0f;
    if (lane < nm) q0 = prow[pm[lane]];
    if (lane + 64 < nm) q1 = prow[pm[lane + 64]];
    if (logits) logits_to_weights(q0, q1, nm);
    if (lane < nm) L.pr[lane] = q0;
    if (lane + 64 < nm) L.pr[lane + 64] = q1;
    wave_sync();
    float all_p = 0.0f;
    if (nm > 0) {
        all_p = L.pr[0];                                   // int 0 + float32
        for (int j = 1; j < nm; ++j) all_p = all_p + L.pr[j];   // float32 accumulation in move order
    }
    if (all_p == 0.0f) all_p = 1.0f;
    if (lane < nm) pp[lane] = L.pr[lane] / all_p;
    if (lane + 64 < nm) pp[lane + 64] = L.pr[lane + 64] / all_p;
    // p[j] is written and later read (select_edge) by the same lane, the header by lane 0: no global fence
    if (lane == 0) store_meta(base, meta & ~(uint32_t)NODE_WAITING);
    wave_sync();
}

// attach_policy + load_path + backup of one evaluated leaf with their independent loads issued together: the move labels
// with the path's edge ids, then the priors of those labels with the statistics of those edges -- two dependent memory
// round trips per leaf instead of five (the BACKUP launch is nothing but such chains, eight leaves one after the other).
// Same arithmetic, same lanes writing the same words as the three functions it replaces.  depth <= 64.
// (round 5) The part of a leaf's chain that does not depend on the tree -- its move labels, its path's edge ids, the network
// row's entries for those labels -- is fetched one leaf AHEAD (LeafPre: leaf_pre_a, then leaf_pre_b once the labels are
// back), under the previous leaf's edge update and fence; what stays in a leaf's serial chain is the edge statistics' load,
// the arithmetic and the store fence: two memory round trips per leaf instead of three.
struct LeafPre {
    uint16_t m0, m1;
    int e;
    float p0, p1;
};
XQ_D void leaf_pre_a(const GameView& gv, LeafPre& lp, int node, uint32_t meta, int depth, const int32_t* __restrict__ hp_edge)
{
    const int lane = lane_id();
    char* base = rec_ptr(gv, (uint32_t)node);
    const int nm = (int)(meta & 0xFF);
    const uint16_t* pm = node_mv(base, nm);
    lp.m0 = 0; lp.m1 = 0; lp.e = 0;
    if (lane < nm) lp.m0 = pm[lane];
    if (lane + 64 < nm) lp.m1 = pm[lane + 64];
    if (lane < depth) lp.e = hp_edge[lane];
}
XQ_D void leaf_pre_b(LeafPre& lp, uint32_t meta, const float* __restrict__ prow)
{
    const int lane = lane_id();
    const int nm = (int)(meta & 0xFF);
    lp.p0 = 0.0f; lp.p1 = 0.0f;
    if (lane < nm) lp.p0 = prow[lp.m0];
    if (lane + 64 < nm) lp.p1 = prow[lp.m1];
}

// `lp`: this leaf's prefetched labels / edge ids / row entries.  `next`: issued between this leaf's edge-statistics load and
// its arithmetic -- the caller's prefetch of the following leaf.
template <typename Next>
XQ_D void attach_and_backup(const SearchParams& P, const GameView& gv, SearchLDS& L, int node, uint32_t meta, int depth,
                            const LeafPre& lp, double v, bool logits, Next&& next)
{
    const int lane = lane_id();
    char* base = rec_ptr(gv, (uint32_t)node);
    const int nm = (int)(meta & 0xFF);
    float* pp = node_p(base);
    float p0 = lp.p0, p1 = lp.p1;
    EdgeStat cur{0.0, 0, 0};
    EdgeStat* ep = nullptr;
    if (lane < depth) { ep = edge_ptr(gv, (uint32_t)lp.e); cur = *ep; }
    next();
    if (logits) logits_to_weights(p0, p1, nm);             // (raw logits: weights relative to the largest legal one)
    // prior spreading (select_action_q_and_u, player.py:272-284): float32 accumulation in move order, the terms read
    // straight from the lanes' registers (a loop over an LDS copy paid an LDS round trip per term)
    float all_p = 0.0f;
    if (nm > 0) {
        all_p = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), 0));          // int 0 + float32
        const int n0 = nm < 64 ? nm : 64;
        for (int j = 1; j < n0; ++j) all_p = all_p + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), j));
        for (int j = 64; j < nm; ++j) all_p = all_p + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), j - 64));
    }
    if (all_p == 0.0f) all_p = 1.0f;
    if (lane < nm) pp[lane] = p0 / all_p;
    if (lane + 64 < nm) pp[lane + 64] = p1 / all_p;
    if (lane == 0) store_meta(base, meta & ~(uint32_t)NODE_WAITING);
    // update_tree (player.py:357-366): lane i = level i
    if (lane < depth) {
        const double vi = ((depth - lane) & 1) ? -v : v;
        ep->n = cur.n + (1 - P.vl);
        ep->w = cur.w + (vi + (double)P.vl);
    }
    count(gv, CT_SIMS);
    count(gv, CT_SUM_DEPTH, (unsigned long long)depth);
    count_max(gv, CT_MAX_DEPTH, (unsigned long long)depth);
    wave_sync_global();
}

// ---- select_action_q_and_u, player.py:286-320 --------------------------------------------------------
struct RootCtx {
    bool is_root;
    int n_no_act;
    const uint16_t* no_act;
    const double* noise;        // this game's Dirichlet rows [K][MAXMOVES] written by k_noise (NULL when eps == 0)
    int sim;
};

struct Picked {
    int j;              // edge index inside the node, -1 = none
    bool have;          // n / w / child / mv below are valid (winner among the first 64 edges)
    int n, child, mv;
    double w;
};

// base: the node's record; sb: its stat block (nullptr: allocated in this visit, every edge is {0, 0, unknown})
XQ_D Picked select_edge(const SearchParams& P, char* base, const EdgeStat* sb, int sum_n, int nm, const RootCtx& rc)
{
    const int lane = lane_id();
    const double xx = __dsqrt_rn((double)(sum_n + 1));
    const float* pp = node_p(base);
    const uint16_t* pm = node_mv(base, nm);
    int n0 = 0, child0 = CHILD_UNKNOWN, mv0 = 0;           // this lane's edge of the first half
    double w0 = 0.0;
    double best_s = -1.0e300;
    int best_j = -1;
    bool win0 = false, win1 = false;
    const int halves = nm > 64 ? 2 : 1;                     // almost every position has <= 64 moves
#pragma nounroll
    for (int h = 0; h < halves; ++h) {
        const int j = lane + 64 * h;
        bool valid = j < nm;
        double score = -1.0e300;
        bool win = false;
        if (valid) {
            EdgeStat es{0.0, 0, CHILD_UNKNOWN};
            if (sb) es = sb[j];                              // N, W and the child link in one 16-byte load
            const float p = pp[j];
            const uint16_t mv = pm[j];
            const int n = es.n;
            const double w = es.w;
            if (h == 0) { n0 = n; w0 = w; child0 = es.child; mv0 = mv; }
            const double q = n ? w / (double)n : 0.0;
            double u;
            if (rc.is_root) {
                if (rc.n_no_act) {
                    for (int k = 0; k < rc.n_no_act; ++k) valid = valid && (rc.no_act[k] != mv);
                }
                const float a = P.one_minus_eps_f32 * p;
                double p_ = (double)a;
                if (rc.noise) p_ = p_ + P.noise_eps * rc.noise[(size_t)rc.sim * MAXMOVES + j];   // player.py:304
                u = P.c_puct * p_ * xx / (double)(1 + n);
            } else {
                const float a = P.c_puct_f32 * p;
                u = (double)a * xx / (double)(1 + n);
            }
            if (valid) {
                score = q + u;
                win = q > (1.0 - 1e-7);
                if (!(score >= -99999999.0)) valid = false;
            }
        }
        if (h == 0) win0 = valid && win; else win1 = valid && win;
        if (valid && score >= best_s) { best_s = score; best_j = j; }   // h = 1 has the larger index: wins ties
    }
    // proven-win shortcut: first edge in order with q > 1 - 1e-7 (player.py:309-311)
    const int first_win = lowest_bit(__ballot(win0), __ballot(win1));
    int pick;
    if (first_win >= 0) pick = first_win;
    else {
        // arg max of (score, index): `>=` keeps the LAST maximal move (player.py:312-314).  The maximum itself by a
        // butterfly on the score alone (never NaN: see `valid`), then the lanes that hold it vote: the largest index wins,
        // and an index of the second half (lane + 64) beats any of the first.
        const double m = wave_max_f64(best_s);                       // DPP ladder (xq_rules.h), no LDS permutes
        const bool top = best_j >= 0 && best_s == m;
        const uint64_t t1 = __ballot(top && best_j >= 64), t0 = __ballot(top);
        pick = t1 ? 64 + (63 - __clzll((long long)t1)) : (t0 ? 63 - __clzll((long long)t0) : -1);
    }
    Picked r;
    r.j = pick;
    r.have = pick >= 0 && pick < 64;
    r.n = 0; r.child = CHILD_UNKNOWN; r.mv = 0; r.w = 0.0;
    if (r.have) {                                              // pick is wave-uniform: register reads, no LDS permute
        r.n = __builtin_amdgcn_readlane(n0, pick);
        r.child = __builtin_amdgcn_readlane(child0, pick);
        r.mv = __builtin_amdgcn_readlane(mv0, pick);
        const long long wb = __double_as_longlong(w0);
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)wb, pick);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)wb >> 32), pick);
        r.w = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return r;
}

// ---- simulation bookkeeping --------------------------------------------------------------------------
XQ_D void sim_finish(const GameView& gv, int sim, int* active)
{
    if (lane_id() == 0) gv.s_state[sim] = SIM_IDLE;
    *active -= 1;
}

// ---- deferred terminal / repetition results -----------------------------------------------------------------------
// MCTS_search does not apply a terminal or repetition value itself: it SUBMITS update_tree to the executor
// (player.py:206, :228-232), behind the search tasks of the batch that were queued first (:173-174), and a search
// thread runs its descent to the end before the interpreter switches.  So the descents of a phase (a fresh batch, or
// the simulations resumed after an evaluation) see each other's virtual losses but not each other's terminal results;
// those are applied when the descents are over, in index order (oracle: finish / flush_deferred
// in xq_mcts.c; tests/golden/kgt1_spread.json is the evidence that this is what the reference does).
// The slot keeps (value, depth) until the flush: written and read back by lane 0 only (its own stores, no fence).
struct Deferred {
    uint64_t lo;       // simulations < 64 waiting for their backup, one bit each (wave-uniform)
    int hi;            // how many with an index >= 64 (found by scanning the slots)
};
XQ_D void defer_result(const GameView& gv, int sim, int depth, int value, Deferred& df)
{
    if (lane_id() == 0) { gv.s_state[sim] = SIM_DEFERRED; gv.s_node[sim] = value; gv.s_depth[sim] = depth; }
    if (sim < 64) df.lo |= 1ull << sim;
    else df.hi += 1;
    wave_sync();
}

XQ_D int find_in_path(const SearchLDS& L, int depth, int node)
{
    const int lane = lane_id();
    for (int base = 0; base < depth; base += 64) {
        const uint64_t m = __ballot(base + lane < depth && L.path_node[base + lane] == node);
        if (m) return base + __ffsll((long long)m) - 1;
    }
    return -1;
}

struct RoundIO {
    void* planes;
    int planes_dtype;
    int in_planes;
    uint32_t* masks;       // cz_search_leaf_masks: [slots][96] occupancy boards, or NULL
    bool planes_off;       // cz_search_leaf_planes(0): only the occupancy boards are written
};

// occupancy-board word of plane position pos (row i = pos / 9 of the planes shows rank y = 9 - i): bit `shift` + plane of the
// piece on that square, 0 for an empty square or pos >= 90 (state_to_planes, static_env.py:137-156)
XQ_D uint32_t mask_word(const int8_t* b, int pos, int shift)
{
    if (pos >= NSQ) return 0u;
    const int i = pos / 9, j = pos - i * 9;
    const int p = b[(9 - i) * 9 + j];
    return p == 0 ? 0u : 1u << (shift + (p > 0 ? p - 1 : 6 - p));
}

XQ_D void encode_block(int dtype, const int8_t* b, uint8_t* codes, char* out)
{
    switch (dtype) {
    case CZ_F32: wave_encode_codes<0>(b, codes, out); break;
    case CZ_F16: wave_encode_codes<1>(b, codes, out); break;
    case CZ_BF16: wave_encode_codes<2>(b, codes, out); break;
    default: wave_encode_codes<3>(b, codes, out); break;
    }
}

// the leaf's input planes into its queue slot: 14 planes (state_to_planes), or 28 with the position two plies
// earlier (`prev`, NULL = none: zeros) as the second block (state_history_to_planes, static_env.py:158-194)
template <bool HIST>
XQ_D void write_planes(const RoundIO& io, const int8_t* b, uint8_t* codes, size_t slot, const int8_t* prev)
{
    const size_t esz = io.planes_dtype == CZ_F32 ? 4 : (io.planes_dtype == CZ_U8 ? 1 : 2);
    char* out = (char*)io.planes + slot * (size_t)io.in_planes * 90 * esz;
    // (round 5) a caller whose network reads the occupancy boards alone (cz_input_resblock_m) switches the planes off
    // (cz_search_leaf_planes): the leaf then costs the code row + two word stores instead of the 1260-element encoder pass
    const bool only_masks = io.masks && io.planes_off;
    if (only_masks) {
        // boards only: one LDS read and a shift per word, straight from the position(s) -- no code row, no encoder pass
        const int lane = lane_id();
        uint32_t m0 = mask_word(b, lane, 0), m1 = mask_word(b, 64 + lane, 0);
        if (HIST && prev) { m0 |= mask_word(prev, lane, 14); m1 |= mask_word(prev, 64 + lane, 14); }
        uint32_t* mo = io.masks + slot * 96;
        mo[lane] = m0;
        if (lane < 32) mo[64 + lane] = m1;
        return;
    }
    encode_block(io.planes_dtype, b, codes, out);
    // the same position as an occupancy board (cz_search_leaf_masks): word pos = plane position i * 9 + j, bit c = plane c shows
    // a piece there -- `codes` holds exactly that channel per position after the encoder's pass
    const int lane = lane_id();
    uint32_t m0 = 0u, m1 = 0u;
    if (io.masks) {
        const uint32_t c0 = codes[lane], c1 = lane < 26 ? codes[64 + lane] : 0xFFu;
        m0 = c0 == 0xFFu ? 0u : 1u << c0;
        m1 = c1 == 0xFFu ? 0u : 1u << c1;
    }
    if (HIST) {
        char* out2 = out + 1260 * esz;
        if (prev) {
            encode_block(io.planes_dtype, prev, codes, out2);
            if (io.masks) {
                const uint32_t c0 = codes[lane], c1 = lane < 26 ? codes[64 + lane] : 0xFFu;
                m0 |= c0 == 0xFFu ? 0u : 1u << (14 + c0);
                m1 |= c1 == 0xFFu ? 0u : 1u << (14 + c1);
            }
        } else {
            for (int q = lane_id(); q < 315 * (int)esz; q += 64) reinterpret_cast<uint32_t*>(out2)[q] = 0u;
        }
    }
    if (io.masks) {
        uint32_t* mo = io.masks + slot * 96;
        mo[lane] = m0;
        if (lane < 32) mo[64 + lane] = m1;
    }
}

// second plane block of a new leaf (expand_and_evaluate, player.py:322-338): simulations started by action() with a
// real game history use the game position two plies before the ROOT whatever their depth (the reference never
// clears `is_root_node` while descending, :217); every other simulation uses its own search path.
XQ_D const int8_t* history_board(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L,
                                 bool fresh, int depth)
{
    if (P.in_planes != 28) return nullptr;
    const int lane = lane_id();
    const int kind = uni((int)B.g_hist_kind[gv.g]);
    if (fresh && kind != 0) {
        if (kind != 1) return nullptr;                                // a history shorter than 5 entries
        const int8_t* pb = B.g_prev_board + (size_t)gv.g * BOARD_LDS;
        L.r.bd[2][lane] = pb[lane];
        if (lane < 32) L.r.bd[2][lane + 64] = (lane < 26) ? pb[lane + 64] : (int8_t)0;
        wave_sync();
        return L.r.bd[2];
    }
    if (depth < 2) return nullptr;
    unpack_key(node_key(gv, L.path_node[depth - 2]), L.r.bd[2]);
    return L.r.bd[2];
}

// The game's heap while a kernel runs: bump pointer + owned chunks, held in registers (wave-uniform).
struct Arena {
    uint32_t top;           // next free granule (record id)
    int nchunks;            // chunks the game owns (reserved by begin_search for the whole ply)
    int ncount;             // nodes in the tree
};

// `granules` <= 128 contiguous granules inside one chunk; 0 = the game's chunks are used up
XQ_D uint32_t heap_alloc(Arena& ar, int granules)
{
    uint32_t top = ar.top;
    if ((int)(top & (uint32_t)(CHUNK_GRANULES - 1)) + granules > CHUNK_GRANULES)
        top = ((top >> CHUNK_SHIFT) + 1u) << CHUNK_SHIFT;            // records never straddle chunks
    if ((int)(top >> CHUNK_SHIFT) >= ar.nchunks) return 0u;
    ar.top = top + (uint32_t)granules;
    return top;
}

// create a node for the position whose packed key is in L.key and whose ordered move list is in `ml` (nm moves);
// returns the node id or -1 when the game's chunks (or its hash table) are full
XQ_D int expand_node(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L,
                     const MoveList& ml, int nm, int hash_slot, uint64_t h, Arena& ar)
{
    const int lane = lane_id();
    nm = nm < MAXMOVES ? nm : MAXMOVES;
    if (hash_slot < 0) return -1;
    const uint32_t id = heap_alloc(ar, NODE_HDR_GRANULES + (6 * nm + 15) / 16);
    if (id == 0u) return -1;
    char* base = rec_ptr(gv, id);
    if (lane < KEY_WORDS) reinterpret_cast<uint32_t*>(base)[lane] = L.key[lane];
    float* pp = node_p(base);
    uint16_t* pm = node_mv(base, nm);
    for (int j = lane; j < nm; j += 64) {
        pp[j] = 0.0f;
        pm[j] = ml.lab[j];
    }
    if (lane == 0) {
        // sum_n = 1 (player.py:213), no statistics yet: they are allocated when the node is first selected from
        *reinterpret_cast<int4*>(base + NODE_OFF_HDR) = make_int4(1, (int)((uint32_t)nm | NODE_WAITING), 0, 0);
        gv.hash[hash_slot] = ((uint64_t)((uint32_t)(h >> 32) | 1u) << 32) | (uint32_t)(id + 1u);
    }
    ar.ncount += 1;
    count(gv, CT_EXPANSIONS);
    count(gv, CT_LEAF_MOVES, (unsigned long long)nm);
    // p / move are written by the lane that reads them in select_edge, the header and the hash slot by lane 0,
    // the key words by the lanes that compare them in hash_lookup: no global fence
    wave_sync();
    return (int)id;
}

// move label of path entry (node, edge): edge - stat block = index into the node's move list (rare: repetitions)
XQ_D int edge_move(const GameView& gv, int node, int edge)
{
    char* base = rec_ptr(gv, (uint32_t)node);
    const NodeHdr hdr = load_hdr(base);
    const int nm = (int)(hdr.meta & 0xFF);
    return uni((int)node_mv(base, nm)[(uint32_t)edge - hdr.stat]);
}

// One descent of simulation `sim` starting at `node` with `depth` path entries already in L.path_*
// (MCTS_search, player.py:198-260).  `node` < 0 means the root position is not in the tree yet.
template <bool HIST>
XQ_D void run_sim(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L,
                  const RoundIO& io, const RootCtx& rc0, int root, int sim, int node, int depth, int* active,
                  Arena& ar, bool fresh, Deferred& df)
{
    const int lane = lane_id();
    const int g = gv.g;
    int32_t* hp_node = gv.path_node + (size_t)sim * P.max_depth;
    int32_t* hp_edge = gv.path_edge + (size_t)sim * P.max_depth;
    if (node < 0) {
        // root expansion (player.py:211-221 with history == [state]); the position is in g_board
        const int8_t* gb = B.g_board + (size_t)g * BOARD_LDS;
        L.r.bd[1][lane] = gb[lane];
        if (lane < 32) L.r.bd[1][lane + 64] = (lane < 26) ? gb[lane + 64] : (int8_t)0;
        wave_sync();
        const int nm = wave_movegen<true>(L.r.bd[1], L.r.ml[0], L.r.plist);
        const uint64_t h = pack_key(L.r.bd[1], L.key);
        int slot;
        int idx = hash_lookup(gv, P, L.key, h, &slot);
        if (idx < 0) idx = expand_node(P, B, gv, L, L.r.ml[0], nm, slot, h, ar);
        if (idx < 0) { count(gv, CT_OVERFLOW_SIMS); backup(P, gv, L, 0, 0.0); sim_finish(gv, sim, active); return; }
        if (lane == 0) {
            B.g_root[g] = idx;
            gv.s_state[sim] = SIM_LEAF; gv.s_node[sim] = idx; gv.s_depth[sim] = 0;
        }
        write_planes<HIST>(io, L.r.bd[1], L.codes, (size_t)g * P.K + sim, HIST ? history_board(P, B, gv, L, fresh, 0) : nullptr);
        wave_sync();
        return;
    }
    PROF_BEGIN();
    for (;;) {
        // state in history[:-1] (player.py:223-236)
        const int rep = find_in_path(L, depth, node);
        if (rep >= 0) {
            unpack_key(node_key(gv, node), L.r.bd[0]);
            const int mv = edge_move(gv, L.path_node[rep], L.path_edge[rep]);
            int v;
            if (wave_will_check_or_catch(L.r, L.r.bd[0], mv) == 1) v = -1;
            else if (wave_be_catched(L.r.bd[0], label_ft(mv) >> 8, L.r.bd[1], L.r.ml[0], L.r.plist)) v = 1;
            else v = 0;
            count(gv, CT_REPETITION_SIMS);
            defer_result(gv, sim, depth, v, df);                     // update_tree is a queued task (player.py:228-232)
            PROF(CT_CYC_REP);
            return;
        }
        char* base = rec_ptr(gv, (uint32_t)node);
        const NodeHdr hdr = load_hdr(base);
        const uint32_t meta = hdr.meta;
        if (meta & NODE_WAITING) {                                  // player.py:238-242
            if (lane == 0) { gv.s_state[sim] = SIM_PARKED; gv.s_node[sim] = node; gv.s_depth[sim] = depth; }
            count(gv, CT_PARKED);
            wave_sync();
            return;
        }
        if (depth >= P.max_depth) {
            count(gv, CT_DEPTH_OVERFLOW);
            backup(P, gv, L, depth, 0.0);
            sim_finish(gv, sim, active);
            return;
        }
        const int nm = (int)(meta & 0xFF);
        // the node's statistics block: allocated on the first selection FROM the node (most nodes stay leaves)
        uint32_t stat = hdr.stat;
        const bool fresh_stat = stat == 0u && nm > 0;
        if (fresh_stat) {
            stat = heap_alloc(ar, nm);
            if (stat == 0u) {
                count(gv, CT_OVERFLOW_SIMS);
                backup(P, gv, L, depth, 0.0);
                sim_finish(gv, sim, active);
                return;
            }
            if (lane == 0) store_stat(base, stat);
            count(gv, CT_STAT_BLOCKS);
        }
        EdgeStat* sb = nm > 0 ? edge_ptr(gv, stat) : nullptr;
        if (fresh_stat) {
            // edge j is initialised by the lane that owns it in select_edge (j & 63): no fence before its later loads
            for (int j = lane; j < nm; j += 64) sb[j] = EdgeStat{0.0, 0, CHILD_UNKNOWN};
        }
        RootCtx rc = rc0;
        rc.is_root = (node == root);                                // player.py:266
        rc.sim = sim;
        const Picked pk = select_edge(P, base, fresh_stat ? nullptr : sb, hdr.sum_n, nm, rc);
        if (pk.j < 0) {                                             // "Best action is None": cannot happen
            backup(P, gv, L, depth, 0.0);
            sim_finish(gv, sim, active);
            return;
        }
        EdgeStat* ep = sb + pk.j;
        const int e = (int)(stat + (uint32_t)pk.j);                 // edge id = granule of its statistics
        const int owner = pk.j & 63;        // the lane that loads this edge in select_edge: its own stores are
                                            // visible to it in program order, no global fence per level
        if (lane == owner) {                                        // player.py:245-252
            if (pk.have) {                                          // the values select just read: no reload
                ep->n = pk.n + P.vl;
                ep->w = pk.w - (double)P.vl;
            } else {
                ep->n += P.vl;
                ep->w = ep->w - (double)P.vl;
            }
        }
        if (lane == 0) {
            store_sum_n(base, hdr.sum_n + 1);
            L.path_node[depth] = node; L.path_edge[depth] = e;
            hp_node[depth] = node; hp_edge[depth] = e;
        }
        count(gv, CT_EDGES_VISITED, (unsigned long long)nm);
        depth += 1;
        wave_sync();
        int child = pk.have ? pk.child : uni(ep->child);
        PROF(CT_CYC_SELECT);
        if (child == CHILD_UNKNOWN) {
            // (the table load of the move's squares is issued before the key's so that the two round trips overlap)
            const int ft = label_ft(pk.have ? pk.mv : uni((int)node_mv(base, nm)[pk.j]));
            unpack_key(reinterpret_cast<const uint32_t*>(base), L.r.bd[0]);
            step_board(L.r.bd[0], ft >> 8, ft & 0xFF, L.r.bd[1]);
            const DoneResult d = wave_done<true>(L.r.bd[1], L.r.bd[2], L.r.ml[0], L.r.ml[1], L.r.plist, false);   // player.py:204
            PROF(CT_CYC_RULES);
            if (d.over) {
                child = d.v > 0 ? CHILD_TERM_WIN : CHILD_TERM_LOSS;
                if (lane == owner) ep->child = child;
            } else {
                const uint64_t h = pack_key(L.r.bd[1], L.key);
                int slot;
                int idx = hash_lookup(gv, P, L.key, h, &slot);
                PROF(CT_CYC_HASH);
                if (idx >= 0) {
                    if (lane == owner) ep->child = idx;
                    child = idx;
                } else {
                    idx = expand_node(P, B, gv, L, L.r.ml[0], d.nmoves, slot, h, ar);     // player.py:211-221
                    if (idx < 0) {
                        count(gv, CT_OVERFLOW_SIMS);
                        backup(P, gv, L, depth, 0.0);
                        sim_finish(gv, sim, active);
                        return;
                    }
                    if (lane == owner) ep->child = idx;
                    if (lane == 0) { gv.s_state[sim] = SIM_LEAF; gv.s_node[sim] = idx; gv.s_depth[sim] = depth; }
                    write_planes<HIST>(io, L.r.bd[1], L.codes, (size_t)g * P.K + sim, HIST ? history_board(P, B, gv, L, fresh, depth) : nullptr);
                    wave_sync();
                    PROF(CT_CYC_EXPAND);
                    return;
                }
            }
            wave_sync();
        }
        if (child == CHILD_TERM_WIN || child == CHILD_TERM_LOSS) {  // player.py:204-208: value doubled
            count(gv, CT_TERMINAL_SIMS);
            defer_result(gv, sim, depth, child == CHILD_TERM_WIN ? 2 : -2, df);     // queued update_tree (player.py:206)
            return;
        }
        node = child;
    }
}

// reload a suspended simulation's path into LDS
XQ_D void load_path(const SearchParams& P, const GameView& gv, SearchLDS& L, int sim, int depth)
{
    const int lane = lane_id();
    const int32_t* hp_node = gv.path_node + (size_t)sim * P.max_depth;
    const int32_t* hp_edge = gv.path_edge + (size_t)sim * P.max_depth;
    wave_sync();
    for (int i = lane; i < depth; i += 64) { L.path_node[i] = hp_node[i]; L.path_edge[i] = hp_edge[i]; }
    wave_sync();
}

// apply the deferred terminal / repetition values of the phase that just ended, in index order
XQ_D void flush_deferred(const SearchParams& P, const GameView& gv, SearchLDS& L, Deferred& df, int* active)
{
    if (df.lo == 0 && df.hi == 0) return;
    wave_sync_global();                  // lane 0's path stores of this launch are complete before the lanes reload paths
    auto one = [&](int i) {
        const int depth = uni(gv.s_depth[i]);
        const double v = (double)uni(gv.s_node[i]);
        load_path(P, gv, L, i, depth);
        backup(P, gv, L, depth, v);
        sim_finish(gv, i, active);
    };
    while (df.lo) {
        const int i = __ffsll((long long)df.lo) - 1;
        df.lo &= df.lo - 1;
        one(i);
    }
    if (df.hi) {
        for (int i = 64; i < P.K; ++i)
            if (uni((int)gv.s_state[i]) == SIM_DEFERRED) one(i);
        df.hi = 0;
    }
}

// ---- the chunk pool ---------------------------------------------------------------------------------------------
// Free chunks sit in a ring: entries [head, tail) are free.  Games take chunks (begin_search) and give them back
// (clear_tree) inside the same kernels, so takers only go up to `tail_vis`, the tail as of an earlier kernel boundary
// (committed by pool_commit): a ring entry is never read in the launch that wrote it.
XQ_D void pool_commit(const SearchBuffers& B)
{
    *B.pool_tail_vis = __hip_atomic_load(B.pool_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one chunk for this game, or -1 when the pool is empty (wave-uniform result)
XQ_D int pool_take(const SearchParams& P, const SearchBuffers& B)
{
    int id = -1;
    if (lane_id() == 0) {
        unsigned int old = __hip_atomic_load(B.pool_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int vis = *B.pool_tail_vis;
        while ((int)(vis - old) > 0) {
            const unsigned int seen = atomicCAS(B.pool_head, old, old + 1u);
            if (seen == old) { id = (int)B.pool_ring[old % (unsigned int)P.n_chunks]; break; }
            old = seen;
        }
    }
    return uni(id);
}

// Drop the game's tree (new game, reset): empty hash table, bump pointer back to the start, every chunk beyond the
// first `keep_chunks` back to the pool.
XQ_D void clear_tree(const SearchParams& P, const SearchBuffers& B, const GameView& gv, uint32_t* chtab)
{
    const int lane = lane_id();
    const int g = gv.g;
    {
        uint4* h4 = reinterpret_cast<uint4*>(gv.hash);                // hash_cap is a power of two >= 64
        for (int i = lane; i < P.hash_cap / 2; i += 64) h4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    const int nch = uni(B.g_nchunks[g]);
    for (int i = P.keep_chunks + lane; i < nch; i += 64) {
        const unsigned int slot = atomicAdd(B.pool_tail, 1u);
        B.pool_ring[slot % (unsigned int)P.n_chunks] = chtab[i];
    }
    if (lane == 0) {
        B.g_nchunks[g] = nch < P.keep_chunks ? nch : P.keep_chunks;
        B.g_heap_top[g] = 1u;                                         // id 0 = "none"
        B.g_node_count[g] = 0;
        B.g_root[g] = -1;
    }
    wave_sync_global();
}

// Make sure the game owns room for `tasks` more simulations (each may add one node record and one statistics block of
// RESERVE_MOVES moves): takes chunks from the pool.  false = the pool, the game's chunk table or its hash table is
// exhausted.
XQ_D bool reserve_ply(const SearchParams& P, const SearchBuffers& B, const GameView& gv, uint32_t* chtab, int tasks)
{
    const int g = gv.g;
    const uint32_t top = uniu(B.g_heap_top[g]);
    int nch = uni(B.g_nchunks[g]);
    const int ncount = uni(B.g_node_count[g]);
    if ((long long)ncount + tasks + 1 > (long long)P.hash_cap * 7 / 8) return false;      // open addressing: keep it sparse
    constexpr int PER_TASK = NODE_HDR_GRANULES + (6 * RESERVE_MOVES + 15) / 16 + RESERVE_MOVES;
    constexpr int USABLE = CHUNK_GRANULES - MAXMOVES;                 // a record that does not fit moves to the next chunk
    const long long need = (long long)(tasks + 1) * PER_TASK;
    long long have = (long long)(nch - (int)(top >> CHUNK_SHIFT)) * USABLE - (long long)(top & (uint32_t)(CHUNK_GRANULES - 1));
    bool ok = true;
    while (have < need) {
        if (nch >= P.max_chunks) { ok = false; break; }
        const int id = pool_take(P, B);
        if (id < 0) { ok = false; break; }
        if (lane_id() == 0) {
            B.g_chunk_tab[(size_t)g * P.max_chunks + nch] = (uint32_t)id;
            chtab[nch] = (uint32_t)id;
        }
        nch += 1;
        have += USABLE;
        count(gv, CT_CHUNKS_TAKEN);
    }
    if (lane_id() == 0) B.g_nchunks[g] = nch;
    wave_sync();
    return ok;
}

// Start the search of the position in g_board (CChessPlayer.action, player.py:145-164): find the
// root in the tree, apply the reuse rule, reserve the ply's memory.
XQ_D void begin_search(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L)
{
    const int lane = lane_id();
    const int g = gv.g;
    const int8_t* gb = B.g_board + (size_t)g * BOARD_LDS;
    L.r.bd[1][lane] = gb[lane];
    if (lane < 32) L.r.bd[1][lane + 64] = (lane < 26) ? gb[lane + 64] : (int8_t)0;
    wave_sync_global();
    const uint64_t h = pack_key(L.r.bd[1], L.key);
    int slot;
    int root = hash_lookup(gv, P, L.key, h, &slot);
    int done_n = root >= 0 ? uni(*reinterpret_cast<const int32_t*>(rec_ptr(gv, (uint32_t)root) + NODE_OFF_HDR)) : 0;   // :153-155
    const int n_no_act = uni((int)B.g_n_no_act[g]), inc = uni((int)B.g_increase_temp[g]);
    if (n_no_act > 0 || inc || done_n == P.sims) done_n = 0;                      // :156-158
    int tasks = P.sims - done_n;
    if (tasks < 0) tasks = 0;
    // Memory policy: the reference keeps a game's tree for the whole game (self_play.py:84,98-100) and so does the
    // engine as long as the pool has chunks.  Only when the coming ply cannot be reserved is the game's tree dropped
    // (counted: tree_resets; the chunks it keeps hold one full search).
    if (tasks > 0 && !reserve_ply(P, B, gv, L.chtab, tasks)) {
        clear_tree(P, B, gv, L.chtab);
        count(gv, CT_TREE_RESETS);
        root = -1;
        tasks = P.sims;
        done_n = 0;
        (void)reserve_ply(P, B, gv, L.chtab, tasks);     // within keep_chunks by construction; else overflow_sims counts
    }
    count(gv, CT_ROOT_REUSED_SIMS, (unsigned long long)done_n);
    if (lane == 0) {
        B.g_root[g] = root;
        B.g_tasks_left[g] = tasks;
        B.g_active[g] = 0;
        B.g_phase[g] = PH_SEARCH;
    }
    wave_sync_global();
}

// ---- calc_policy + apply_temperature + choice (player.py:375-406, 453-470, :195) ---------------------
// returns the chosen label, or -1 when the player resigns.  u is the uniform draw of np.random.choice.
XQ_D int choose_action(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L, double u,
                       bool enable_resign)
{
    const int lane = lane_id();
    const int g = gv.g;
    const int root = uni(B.g_root[g]);
    if (root < 0) return -2;
    char* base = rec_ptr(gv, (uint32_t)root);
    const NodeHdr hdr = load_hdr(base);
    const int nm = (int)(hdr.meta & 0xFF);
    if (nm == 0) return -1;             // no move at all (the reference player dead-locks here): give the game up
    const uint16_t* pm = node_mv(base, nm);
    const EdgeStat* sb = hdr.stat ? edge_ptr(gv, hdr.stat) : nullptr;   // none: the root was never selected from
    const int n_no_act = uni((int)B.g_n_no_act[g]);
    const uint16_t* no_act = B.g_no_act + (size_t)g * MAX_NO_ACT;
    const int turns = uni(B.g_turns[g]);
    // visit counts (banned -> 0), max q over the non-banned edges
    int cnt[2];
    double maxq = -100.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        cnt[h] = 0;
        if (j < nm) {
            EdgeStat es{0.0, 0, CHILD_UNKNOWN};
            if (sb) es = sb[j];
            const int n = es.n;
            const uint16_t mv = pm[j];
            bool banned = false;
            for (int k = 0; k < n_no_act; ++k) banned = banned || (no_act[k] == mv);
            if (!banned) {
                cnt[h] = n;
                const double q = n ? es.w / (double)n : 0.0;
                maxq = q > maxq ? q : maxq;
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_xor(maxq, d, 64);
        maxq = o > maxq ? o : maxq;
    }
    maxq = unid(maxq);
    if (maxq < P.resign_threshold && enable_resign && turns > P.min_resign_turn) return -1;   // :397-398
    // order the edges by label (the policy vector is indexed by label)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = lane + 64 * h;
        if (j < nm) {
            const uint16_t mv = pm[j];
            int rank = 0;
            for (int k = 0; k < nm; ++k) rank += pm[k] < mv;
            L.slab[rank] = mv;
            L.sn[rank] = cnt[h];
        }
    }
    wave_sync_global();
    // temperature (player.py:453-461)
    const int inc = uni((int)B.g_increase_temp[g]);
    double tau = 0.0;
    if (turns < 30 && P.tau_decay_rate != 0.0) tau = pow(P.tau_decay_rate, (double)(turns + 1));
    if (tau < 0.1 || (turns >= 4 && P.evaluate)) tau = 0.0;
    if (inc && !P.evaluate) tau = 0.5;
    int chosen = 0;
    if (tau == 0.0) {
        if (lane == 0) {
            int best = -1, bestn = -1;                       // np.argmax: first maximum in label order
            for (int k = 0; k < nm; ++k) if (L.sn[k] > bestn) { bestn = L.sn[k]; best = k; }
            chosen = (best >= 0 && bestn > 0) ? (int)L.slab[best] : 0;
        }
        wave_sync_global();
        return uni(chosen);
    }
    // policy ** (1 / tau), renormalised, then np.random.choice = cdf.searchsorted(u, side='right').  Everything that is
    // independent per move (the pow() calls above all: ~1 us each when one lane runs them back to back) is done one move
    // per lane; the float64 sums keep the order of the NumPy loops and stay on lane 0.
    long long total_n = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) if (lane + 64 * h < nm) total_n += L.sn[lane + 64 * h];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) total_n += __shfl_xor(total_n, d, 64);     // integers: any order
    const double inv = 1.0 / tau;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k < nm) {
            const double pk = (double)L.sn[k] / (double)total_n;             // policy /= np.sum(policy)
            L.dsc[k] = L.sn[k] > 0 ? pow(pk, inv) : 0.0;
        }
    }
    wave_sync_global();
    double s = 0.0;
    if (lane == 0) for (int k = 0; k < nm; ++k) s += L.dsc[k];
    s = unid(s);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k < nm) L.dsc[k] = L.dsc[k] / s;
    }
    wave_sync_global();
    double total = 0.0;
    if (lane == 0) {
        double c = 0.0;
        for (int k = 0; k < nm; ++k) { total += L.dsc[k]; c += L.dsc[k]; L.cdf[k] = c; }
    }
    total = unid(total);
    wave_sync_global();
    bool hit[2], pos[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        hit[h] = k < nm && L.cdf[k] / total > u;
        pos[h] = k < nm && L.dsc[k] > 0.0;
    }
    int pick = lowest_bit(__ballot(hit[0]), __ballot(hit[1]));          // first k with cdf[k] / total > u
    if (pick < 0) {                                                      // u beyond the last step: the last possible move
        const uint64_t p0 = __ballot(pos[0]), p1 = __ballot(pos[1]);
        pick = p1 ? 127 - __clzll((long long)p1) : (p0 ? 63 - __clzll((long long)p0) : -1);
    }
    chosen = pick >= 0 ? (int)L.slab[pick] : 0;
    wave_sync_global();
    return uni(chosen);
}

XQ_D void store_key(uint32_t* dst, const uint32_t* key)
{
    const int lane = lane_id();
    if (lane < KEY_WORDS) dst[lane] = key[lane];
}

XQ_D void set_init_board(int8_t* gb)
{
    // rkemsmekr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RKEMSMEKR (static_env.py:9)
    const int lane = lane_id();
    const int8_t back[9] = {ROOK, KNIGHT, ELEPHANT, ADVISOR, KING, ADVISOR, ELEPHANT, KNIGHT, ROOK};
    for (int s = lane; s < BOARD_LDS; s += 64) {
        int p = 0;
        if (s < NSQ) {
            const int x = s % 9, y = s / 9;
            if (y == 0) p = back[x];
            else if (y == 9) p = -back[x];
            else if (y == 2 && (x == 1 || x == 7)) p = CANNON;
            else if (y == 7 && (x == 1 || x == 7)) p = -CANNON;
            else if (y == 3 && (x % 2 == 0)) p = PAWN;
            else if (y == 6 && (x % 2 == 0)) p = -PAWN;
        }
        gb[s] = (int8_t)p;
    }
}

// (re)start a self-play game in slot g (SelfPlayWorker.start_game, self_play.py:95-116)
XQ_D void new_game(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L, uint32_t game_id)
{
    const int lane = lane_id();
    const int g = gv.g;
    clear_tree(P, B, gv, L.chtab);
    set_init_board(B.g_board + (size_t)g * BOARD_LDS);
    wave_sync_global();
    const int8_t* gb = B.g_board + (size_t)g * BOARD_LDS;
    L.r.bd[1][lane] = gb[lane];
    if (lane < 32) L.r.bd[1][lane + 64] = gb[lane + 64];
    wave_sync_global();
    const uint64_t h0 = pack_key(L.r.bd[1], L.key);
    store_key(B.g_hist_key + (size_t)g * (P.max_plies + 2) * KEY_WORDS, L.key);
    if (lane == 0) B.g_hist_hash[(size_t)g * (P.max_plies + 2)] = h0;
    if (lane == 0) {
        B.g_game_id[g] = game_id;
        B.g_turns[g] = 0;
        B.g_no_eat[g] = 0;
        B.g_n_no_act[g] = 0;
        B.g_increase_temp[g] = 0;
        // enable_resign = random() > enable_resign_rate (self_play.py:102-105): stream 0, draw 0
        B.g_enable_resign[g] = philox_uniform(P.seed, game_id, 0, 0) > P.enable_resign_rate ? 1 : 0;
    }
    wave_sync_global();
    begin_search(P, B, gv, L);
}

XQ_D void emit_record(const SearchParams& P, const SearchBuffers& B, const GameView& gv, int turns, int value,
                      bool store, bool resigned)
{
    const int lane = lane_id();
    const int g = gv.g;
    unsigned int pos = 0;
    if (lane == 0) pos = atomicAdd(B.ring_tail, 1u);
    pos = uniu(pos);
    uint8_t* rec = B.ring + (size_t)(pos % (unsigned)P.ring_cap) * P.record_stride;
    if (lane == 0) {
        GameRecord* hdr = reinterpret_cast<GameRecord*>(rec);
        hdr->game_id = B.g_game_id[g];
        hdr->turns = turns;
        hdr->value = value;
        hdr->flags = (store ? 1u : 0u) | (resigned ? 2u : 0u);
    }
    uint16_t* mv = reinterpret_cast<uint16_t*>(rec + sizeof(GameRecord));
    const uint16_t* acts = B.g_hist_act + (size_t)g * (P.max_plies + 2);
    for (int i = lane; i < turns && i < P.max_plies + 2; i += 64) mv[i] = acts[i];
    count(gv, CT_GAMES);
    count(gv, value > 0 ? CT_RED_WINS : (value < 0 ? CT_BLACK_WINS : CT_DRAWS));
    if (resigned) count(gv, CT_RESIGNS);
}

// One ply of SelfPlayWorker.start_game after the search finished (self_play.py:124-212).
XQ_D void advance_game(const SearchParams& P, const SearchBuffers& B, const GameView& gv, SearchLDS& L)
{
    const int lane = lane_id();
    const int g = gv.g;
    const uint32_t game_id = uniu(B.g_game_id[g]);
    int turns = uni(B.g_turns[g]);
    int8_t* gb = B.g_board + (size_t)g * BOARD_LDS;
    uint32_t* hkeys = B.g_hist_key + (size_t)g * (P.max_plies + 2) * KEY_WORDS;
    uint16_t* hacts = B.g_hist_act + (size_t)g * (P.max_plies + 2);
    const double u = philox_uniform(P.seed, game_id, 1, (uint64_t)turns);
    const int action = choose_action(P, B, gv, L, u, uni((int)B.g_enable_resign[g]) != 0);
    count(gv, CT_PLIES);
    bool game_over = false, resigned = false;
    int value = 0;
    int final_move = NOMOVE;
    if (action < 0) {                                                  // resign, :126-129
        value = -1; game_over = true; resigned = true;
    } else {
        // state, no_eat = senv.new_step(state, action)  (:133-141)
        L.r.bd[0][lane] = gb[lane];
        if (lane < 32) L.r.bd[0][lane + 64] = (lane < 26) ? gb[lane + 64] : (int8_t)0;
        wave_sync_global();
        const int ft = label_ft(action);
        const int f = ft >> 8, t = ft & 0xFF;
        const bool no_eat = L.r.bd[0][t] == 0;
        step_board(L.r.bd[0], f, t, L.r.bd[3]);                         // bd[3] = new state, kept below
        if (lane == 0) hacts[turns] = (uint16_t)action;
        turns += 1;
        int no_eat_count = uni(B.g_no_eat[g]);
        no_eat_count = no_eat ? no_eat_count + 1 : 0;
        const uint64_t h = pack_key(L.r.bd[3], L.key);
        uint64_t* hhash = B.g_hist_hash + (size_t)g * (P.max_plies + 2);
        if (turns < P.max_plies + 2) {
            store_key(hkeys + (size_t)turns * KEY_WORDS, L.key);
            if (lane == 0) hhash[turns] = h;
        }
        int n_no_act = 0, inc = 0;
        if (no_eat_count >= 120 || turns >= 2 * P.max_game_length) {    // :149-151
            game_over = true; value = 0;
        } else {
            const DoneResult d = wave_done<true>(L.r.bd[3], L.r.bd[1], L.r.ml[0], L.r.ml[1], L.r.plist, true);   // :153
            game_over = d.over != 0; value = d.v; final_move = d.final_move;
            if (!game_over && !wave_has_attack(L.r.bd[3])) { game_over = true; value = 0; }  // :154-158
            if (!game_over && !d.check) {                               // :161-175
                int free_move = 0;
                // earlier states equal to this one, in order (both parities: the reference compares strings)
                // (64 earlier plies per step: lane l compares the 64-bit hash of ply base + l, one ballot lists the
                //  candidates, which are then verified word by word; a ply-at-a-time scan was 200 dependent memory
                //  round trips per move late in a game)
                for (int base = 0; base < turns && !game_over; base += 64) {
                  const int cmp_ply = base + lane;
                  uint64_t cand = __ballot(cmp_ply < turns && hhash[cmp_ply] == h);
                  while (cand && !game_over) {
                    const int i = base + __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    const bool differs = lane < KEY_WORDS && hkeys[(size_t)i * KEY_WORDS + lane] != L.key[lane];
                    if (__ballot(differs)) continue;                  // a hash collision
                    const int mv = uni((int)hacts[i]);
                    // the rule helpers need the position in bd[0]
                    L.r.bd[0][lane] = L.r.bd[3][lane];
                    if (lane < 32) L.r.bd[0][lane + 64] = L.r.bd[3][lane + 64];
                    wave_sync_global();
                    if (wave_will_check_or_catch(L.r, L.r.bd[0], mv) == 1) {
                        if (n_no_act < MAX_NO_ACT) {
                            if (lane == 0) B.g_no_act[(size_t)g * MAX_NO_ACT + n_no_act] = (uint16_t)mv;
                            n_no_act += 1;
                        } else {
                            count(gv, CT_NO_ACT_TRUNCATED);           // (the reference's list is unbounded, self_play.py:161-175)
                        }
                    } else if (!wave_be_catched(L.r.bd[0], label_ft(mv) >> 8, L.r.bd[1], L.r.ml[0], L.r.plist)) {
                        inc = 1;
                        free_move += 1;
                        if (free_move >= 3) { game_over = true; value = 0; }
                    }
                  }
                }
            }
        }
        // commit the new position
        gb[lane] = L.r.bd[3][lane];
        if (lane < 32) gb[lane + 64] = (lane < 26) ? L.r.bd[3][lane + 64] : (int8_t)0;
        if (lane == 0) {
            B.g_turns[g] = turns;
            B.g_no_eat[g] = no_eat_count;
            B.g_n_no_act[g] = (uint8_t)n_no_act;
            B.g_increase_temp[g] = (uint8_t)inc;
        }
        wave_sync_global();
    }
    if (!game_over) {
        begin_search(P, B, gv, L);
        return;
    }
    if (final_move != NOMOVE) {                                         // :177-184
        if (lane == 0 && turns < P.max_plies + 2) hacts[turns] = (uint16_t)final_move;
        turns += 1;
        value = -value;
    }
    if (turns % 2 == 1) value = -value;                                 // :190-191
    bool store = true;
    if (turns < 10) store = philox_uniform(P.seed, game_id, 0, 1) > 0.9;   // :194-200
    wave_sync_global();
    emit_record(P, B, gv, turns, value, store, resigned);
    new_game(P, B, gv, L, game_id + P.game_id_stride);
}

// ---- the round kernels --------------------------------------------------------------------------------------
// One lock-step round = k_sim(BACKUP) -> k_advance -> k_sim(SELECT)   (+ k_noise before k_sim(SELECT) when the
// root noise is on).  Splitting the round keeps the hot simulation kernel free of the cold, register-hungry code
// (move sampling with pow(), game rules, chunk reservation, Gamma sampling).
constexpr int SIM_BACKUP = 1, SIM_SELECT = 2;

template <bool HIST>        // HIST: 28 input planes (use_history); kept out of the common 14-plane instantiation
__global__ __launch_bounds__(64, 4) void k_sim(SearchParams P, SearchBuffers B, const float* __restrict__ policy,
                                           const float* __restrict__ value, void* planes, int mask, int compact,
                                           int32_t* __restrict__ q_rows, int32_t* __restrict__ q_count)
{
    __shared__ SearchLDS L;
    const int g = blockIdx.x;
    // first kernel of a round: chunks returned during the previous round become takeable (see pool_commit)
    if (g == 0 && (mask & SIM_BACKUP) && lane_id() == 0) {
        pool_commit(B);
        if (q_count) *q_count = 0;          // compact queue of this round: filled by the k_sim(SELECT) launch below
    }
    if (g >= P.G) return;
    if (uni((int)B.g_phase[g]) != PH_SEARCH) return;
    const GameView gv = make_view(B, P, g, L.ctr, L.chtab);
    counters_begin(gv);
#ifdef CZ_SIM_PROFILE
    const long long prof_k0 = clock64();
#endif
    const RoundIO io{planes, P.planes_dtype, P.in_planes, B.leaf_masks, B.leaf_planes_off != 0};
    int active = uni(B.g_active[g]);
    Arena ar{uniu(B.g_heap_top[g]), uni(B.g_nchunks[g]), uni(B.g_node_count[g])};
    const RootCtx rc{false, uni((int)B.g_n_no_act[g]), B.g_no_act + (size_t)g * MAX_NO_ACT,
                     P.noise_eps != 0.0 ? B.noise + (size_t)g * P.K * MAXMOVES : nullptr, 0};
    int resume_i = P.K;
    // the slot table as it is when the launch starts, slot i on lane i: one load per array instead of one dependent
    // round trip per slot and field (a simulation only ever changes its own slot, and each slot is visited once per list)
    const bool snap = (mask & SIM_BACKUP) && active > 0 && P.K <= 64;
    int sn_state = SIM_IDLE, sn_node = 0, sn_depth = 0;
    if (snap && lane_id() < P.K) {
        sn_state = gv.s_state[lane_id()];
        sn_node = gv.s_node[lane_id()];
        sn_depth = gv.s_depth[lane_id()];
    }
    if (snap) {
        // 1. attach + backup evaluated leaves, in simulation order (update_tree, player.py:340-373)
        const int lane = lane_id();
        int my_row = g * P.K + lane;
        uint32_t my_meta = 0u;
        float my_v = 0.0f;
        if (sn_state == SIM_LEAF) {
            if (compact) {                                            // the previous round built a compact queue: the row
                const int row = B.s_qrow[(size_t)g * P.K + lane];     // it gave this leaf
                if (row >= 0) my_row = row;
            }
            my_meta = *reinterpret_cast<const uint32_t*>(rec_ptr(gv, (uint32_t)sn_node) + NODE_OFF_HDR + 4);
            my_v = value[my_row];
        }
        PROF_BEGIN();
        // the evaluated leaves of this game, in simulation order, as a bit set; leaf_at(i) = its wave-uniform fields
        const uint64_t leaves = __ballot(sn_state == SIM_LEAF) & (P.K < 64 ? (1ull << P.K) - 1ull : ~0ull);
        auto pre_a = [&](LeafPre& lp, int i) {
            const int depth = __builtin_amdgcn_readlane(sn_depth, i);
            if (depth <= 64)
                leaf_pre_a(gv, lp, __builtin_amdgcn_readlane(sn_node, i), (uint32_t)__builtin_amdgcn_readlane((int)my_meta, i),
                           depth, gv.path_edge + (size_t)i * P.max_depth);
        };
        auto pre_b = [&](LeafPre& lp, int i) {
            if (__builtin_amdgcn_readlane(sn_depth, i) <= 64)
                leaf_pre_b(lp, (uint32_t)__builtin_amdgcn_readlane((int)my_meta, i),
                           policy + (size_t)__builtin_amdgcn_readlane(my_row, i) * NLABELS);
        };
        LeafPre lp{0, 0, 0, 0.0f, 0.0f};
        uint64_t rest = leaves;
        if (rest) {
            const int i0 = __ffsll((long long)rest) - 1;
            pre_a(lp, i0);
            pre_b(lp, i0);
        }
        while (rest) {
            const int i = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            const int i_next = rest ? __ffsll((long long)rest) - 1 : -1;
            const int node = __builtin_amdgcn_readlane(sn_node, i);
            const int depth = __builtin_amdgcn_readlane(sn_depth, i);
            const size_t slot = (size_t)__builtin_amdgcn_readlane(my_row, i);
            const double v = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), i));   // float(v) of a float32
            LeafPre nx{0, 0, 0, 0.0f, 0.0f};
            if (depth <= 64) {
                attach_and_backup(P, gv, L, node, (uint32_t)__builtin_amdgcn_readlane((int)my_meta, i), depth, lp, v,
                                  P.policy_logits != 0, [&]() {
                                      if (i_next >= 0) { pre_a(nx, i_next); pre_b(nx, i_next); }
                                  });
            } else {
                attach_policy(gv, L, node, policy + slot * NLABELS, P.policy_logits != 0);
                load_path(P, gv, L, i, depth);
                backup(P, gv, L, depth, v);
                if (i_next >= 0) { pre_a(nx, i_next); pre_b(nx, i_next); }
            }
            lp = nx;
            sim_finish(gv, i, &active);
        }
        PROF(CT_CYC_ATTACH);
        resume_i = 0;                                                 // 2. then resume the parked simulations
    } else if ((mask & SIM_BACKUP) && active > 0) {                   // (more than 64 slots per game: slot by slot)
        for (int i = 0; i < P.K; ++i) {
            if (uni((int)gv.s_state[i]) != SIM_LEAF) continue;
            const int node = uni(gv.s_node[i]);
            const int depth = uni(gv.s_depth[i]);
            size_t slot = (size_t)g * P.K + i;
            if (compact) {
                const int row = uni(B.s_qrow[slot]);
                if (row >= 0) slot = (size_t)row;
            }
            attach_policy(gv, L, node, policy + slot * NLABELS, P.policy_logits != 0);
            load_path(P, gv, L, i, depth);
            backup(P, gv, L, depth, (double)value[slot]);             // float(v) of a float32
            sim_finish(gv, i, &active);
        }
        resume_i = 0;
    }
    int new_i = 0, new_n = 0, batches = 0;
    Deferred df{0ull, 0};
    for (int guard = 0; guard < (1 << 20); ++guard) {
        int sim, node, depth;
        bool fresh;
        if (resume_i < P.K) {                                         // parked simulations, in index order
            const int i = resume_i++;
            if (snap) {
                if (__builtin_amdgcn_readlane(sn_state, i) != SIM_PARKED) continue;
                node = __builtin_amdgcn_readlane(sn_node, i); depth = __builtin_amdgcn_readlane(sn_depth, i);
            } else {
                if (uni((int)gv.s_state[i]) != SIM_PARKED) continue;
                node = uni(gv.s_node[i]); depth = uni(gv.s_depth[i]);
            }
            sim = i; fresh = false;
            load_path(P, gv, L, i, depth);
        } else if (new_i < new_n) {                                   // the simulations of a fresh batch
            sim = new_i++; node = uni(B.g_root[g]); depth = 0; fresh = true;
        } else if (df.lo != 0 || df.hi != 0) {
            // the descents of this phase (resumed simulations / a fresh batch) are over: the terminal and repetition
            // values they produced are applied now, in index order
            flush_deferred(P, gv, L, df, &active);
            continue;
        } else if ((mask & SIM_SELECT) && active == 0) {              // 3. next lock-step batch (player.py:169-178)
            const int tasks = uni(B.g_tasks_left[g]);
            if (tasks <= 0) break;                                    // search complete: k_advance takes over
            // A batch whose simulations all end on terminal / repeated positions needs no evaluation and the next one
            // could start at once -- in a won endgame that chains hundreds of batches inside one launch and the whole
            // round waits for this wave.  The order of simulations does not depend on where the chain is cut.
            if (++batches > P.max_batches) break;
            new_n = tasks < P.K ? tasks : P.K;
            new_i = 0;
            active = new_n;
            if (lane_id() == 0) B.g_tasks_left[g] = tasks - new_n;
            continue;
        } else break;
        run_sim<HIST>(P, B, gv, L, io, rc, uni(B.g_root[g]), sim, node, depth, &active, ar, fresh, df);
    }
    if (lane_id() == 0) { B.g_active[g] = active; B.g_node_count[g] = ar.ncount; B.g_heap_top[g] = ar.top; }
    if ((mask & SIM_SELECT) && q_rows) {
        // Compact evaluation queue: this game's slots that hold a new leaf (also those a resumed simulation made in
        // the BACKUP launch) are appended to q_rows -- one atomic per game, the order of the games is arbitrary -- and
        // each remembers its row for the next round's attach (s_qrow).
        wave_sync_global();                                   // lane 0 wrote the slot states
        const int lane = lane_id();
        for (int base = 0; base < P.K; base += 64) {
            const int i = base + lane;
            const bool leaf = i < P.K && gv.s_state[i] == SIM_LEAF;
            const uint64_t m = __ballot(leaf);
            if (m == 0) continue;
            int at = 0;
            if (lane == 0) at = atomicAdd(q_count, __popcll(m));
            at = uni(at);
            if (leaf) {
                const int row = at + __popcll(m & ((1ull << lane) - 1ull));
                q_rows[row] = g * P.K + i;
                B.s_qrow[(size_t)g * P.K + i] = row;
            }
        }
    }
#ifdef CZ_SIM_PROFILE
    count(gv, (mask & SIM_SELECT) ? CT_CYC_KERNEL_SELECT : CT_CYC_KERNEL_BACKUP, (unsigned long long)(clock64() - prof_k0));
#endif
    counters_flush(gv);
}

// End of a search: external mode marks the game READY; self-play mode plays the move, applies the game rules
// and sets up the next search (or the next game).
__global__ __launch_bounds__(64) void k_advance(SearchParams P, SearchBuffers B)
{
    __shared__ SearchLDS L;
    const int g = blockIdx.x;
    if (g >= P.G) return;
    if (uni((int)B.g_phase[g]) != PH_SEARCH || uni(B.g_active[g]) != 0 || uni(B.g_tasks_left[g]) != 0) return;
    const GameView gv = make_view(B, P, g, L.ctr, L.chtab);
    counters_begin(gv);
    if (P.mode == MODE_SELFPLAY) {
        for (int it = 0; it < 8; ++it) {          // a search with nothing to do (fully reused root) ends at once
            advance_game(P, B, gv, L);
            if (uni((int)B.g_phase[g]) != PH_SEARCH || uni(B.g_tasks_left[g]) != 0) break;
        }
    } else if (lane_id() == 0) {
        B.g_phase[g] = PH_READY;
    }
    counters_flush(gv);
}

// Dirichlet(alpha 1_n)[0] for every (simulation slot, root edge) of the batch that the next k_sim(SELECT) launch starts:
// X / (X + Y), X ~ Gamma(alpha), Y ~ Gamma(alpha (n - 1)) (xq_noise.h).  The reference redraws it per move per root visit
// (player.py:304); a simulation selects at the root at most once, so one row per slot per batch is enough.
// ONE launch per round, before k_sim(SELECT), and only games that start a batch there draw anything.  The rare game
// whose root is not in the tree yet (first search of a game, a search after a reset) needs rows too: its simulation 0
// expands the root, the others park on it and select at the root when the NEXT round's k_sim(BACKUP) resumes them --
// with the rows drawn here, for the move count this kernel works out itself from the root position (round 2 ran a
// second k_noise launch before every k_sim(BACKUP) for that case: 19 us per round for 4096 waves that found nothing).
// (Drawing the rows inside k_sim, by the game's own wave when a batch starts, was measured and is SLOWER: the 6 x 64
// draws of a batch are dependent instruction chains for one wave; here they spread over 4 waves per game.)
__global__ __launch_bounds__(256) void k_noise(SearchParams P, SearchBuffers B)
{
    __shared__ int s_nm;
    const int g = blockIdx.x;
    if (g >= P.G || B.g_phase[g] != PH_SEARCH) return;
    // a new batch starts only when nothing is in flight (k_sim(BACKUP) may just have finished the old one)
    if (B.g_active[g] != 0) return;
    const int tasks = B.g_tasks_left[g];
    const int last = tasks < P.K ? tasks : P.K;                 // slots [0, last)
    if (last <= 0) return;
    const int root = B.g_root[g];
    const int tid = threadIdx.x;
    int nm;
    if (root >= 0) {
        const char* rbase = B.pool + ((size_t)B.g_chunk_tab[(size_t)g * P.max_chunks + ((uint32_t)root >> CHUNK_SHIFT)] << 20)
                            + ((size_t)((uint32_t)root & (uint32_t)(CHUNK_GRANULES - 1)) << 4);
        nm = (int)(*reinterpret_cast<const uint32_t*>(rbase + NODE_OFF_HDR + 4) & 0xFF);
    } else {
        // the root position is in g_board: count its moves from wave_movegen's three ballot sets (gen_piece's counting
        // form, one or two squares per lane), without the lists.  Wave 0 only; the other waves wait at the barrier.
        if (tid < 64) {
            const int8_t* gb = B.g_board + (size_t)g * BOARD_LDS;
            const int p0 = gb[tid];
            const int p1 = tid < 26 ? gb[tid + 64] : 0;
            const Set90 occ{__ballot(p0 != 0), __ballot(p1 != 0)};
            const Set90 own{__ballot(p0 > 0), __ballot(p1 > 0)};
            const Set90 oking{__ballot(p0 == -KING), __ballot(p1 == -KING)};
            int c = 0;
            if (p0 > 0) c += gen_piece<false>(p0, tid, occ, own, oking, nullptr, nullptr, 0);
            if (p1 > 0) c += gen_piece<false>(p1, tid + 64, occ, own, oking, nullptr, nullptr, 0);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
            if (tid == 0) s_nm = c < MAXMOVES ? c : MAXMOVES;      // (expand_node caps the list the same way)
        }
        __syncthreads();
        nm = s_nm;
    }
    const uint32_t epoch = B.g_noise_epoch[g];
    double* rows = B.noise + (size_t)g * P.K * MAXMOVES;
    const float alpha = (float)P.dirichlet_alpha;
    const uint32_t key = B.g_game_id[g] + (uint32_t)g * 2654435761u;
    // one (simulation slot, root move) pair per thread and step: 8 x 44 pairs are two steps of the 256 threads
    for (int item = tid; item < last * nm; item += (int)blockDim.x) {
        const int sim = item / nm, j = item - sim * nm;
        NoiseRng rng = NoiseRng::make(P.seed, key, epoch, (uint32_t)sim, (uint32_t)j);
        rows[(size_t)sim * MAXMOVES + j] = dirichlet0(alpha, nm, rng);
    }
    __syncthreads();
    if (tid == 0) B.g_noise_epoch[g] = epoch + 1;
}

// ---- auxiliary kernels ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_start_selfplay(SearchParams P, SearchBuffers B, uint32_t first_game_id)
{
    __shared__ SearchLDS L;
    const int g = blockIdx.x;
    if (g >= P.G) return;
    const GameView gv = make_view(B, P, g, B.counters + (size_t)g * CT_COUNT, L.chtab);   // counts go straight to the global block
    const int lane = lane_id();
    for (int i = lane; i < P.K; i += 64) gv.s_state[i] = SIM_IDLE;
    for (int i = lane; i < CT_COUNT; i += 64) gv.ctr[i] = 0;
    wave_sync_global();
    new_game(P, B, gv, L, first_game_id + (uint32_t)g);
}

__global__ __launch_bounds__(64) void k_set_roots(SearchParams P, SearchBuffers B, const int8_t* __restrict__ boards,
                                                 const int32_t* __restrict__ turns, const uint16_t* __restrict__ no_act,
                                                 const uint8_t* __restrict__ n_no_act, const uint8_t* __restrict__ inc,
                                                 const uint8_t* __restrict__ enable_resign,
                                                 const uint8_t* __restrict__ select_mask,
                                                 const int8_t* __restrict__ prev_boards,
                                                 const uint8_t* __restrict__ hist_kind)
{
    __shared__ SearchLDS L;
    const int g = blockIdx.x;
    if (g >= P.G) return;
    if (select_mask && !select_mask[g]) return;
    const GameView gv = make_view(B, P, g, B.counters + (size_t)g * CT_COUNT, L.chtab);   // counts go straight to the global block
    const int lane = lane_id();
    load_board(boards + (size_t)g * NSQ, L.r.bd[0]);
    int8_t* gb = B.g_board + (size_t)g * BOARD_LDS;
    gb[lane] = L.r.bd[0][lane];
    if (lane < 32) gb[lane + 64] = L.r.bd[0][lane + 64];
    const int nna = n_no_act ? (int)n_no_act[g] : 0;
    if (lane < MAX_NO_ACT) B.g_no_act[(size_t)g * MAX_NO_ACT + lane] = (no_act && lane < nna) ? no_act[(size_t)g * MAX_NO_ACT + lane] : (uint16_t)NOMOVE;
    for (int i = lane; i < P.K; i += 64) gv.s_state[i] = SIM_IDLE;
    {
        const int hk = hist_kind ? (int)hist_kind[g] : 0;
        int8_t* pb = B.g_prev_board + (size_t)g * BOARD_LDS;
        if (hk == 1 && prev_boards) {
            for (int i = lane; i < NSQ; i += 64) pb[i] = prev_boards[(size_t)g * NSQ + i];
        }
        if (lane == 0) B.g_hist_kind[g] = (uint8_t)((hk == 1 && !prev_boards) ? 2 : hk);
    }
    if (lane == 0) {
        B.g_turns[g] = turns ? turns[g] : 0;
        B.g_n_no_act[g] = (uint8_t)(nna < MAX_NO_ACT ? nna : MAX_NO_ACT);
        B.g_increase_temp[g] = inc ? inc[g] : 0;
        B.g_enable_resign[g] = enable_resign ? enable_resign[g] : 0;
    }
    wave_sync_global();
    // a terminal root has nothing to search (the reference's simulations would all return at once)
    const DoneResult d = wave_done(L.r.bd[0], L.r.bd[1], L.r.ml[0], L.r.ml[1], L.r.plist, false);
    begin_search(P, B, gv, L);
    if (d.over && lane == 0) { B.g_tasks_left[g] = 0; }
}

__global__ __launch_bounds__(64) void k_reset_trees(SearchParams P, SearchBuffers B)
{
    __shared__ uint32_t chtab[MAX_CHUNKS];
    const int g = blockIdx.x;
    if (g >= P.G) return;
    const GameView gv = make_view(B, P, g, B.counters + (size_t)g * CT_COUNT, chtab);   // counts go straight to the global block
    clear_tree(P, B, gv, chtab);
    const int lane = lane_id();
    for (int i = lane; i < P.K; i += 64) gv.s_state[i] = SIM_IDLE;
    if (lane == 0) { B.g_phase[g] = PH_IDLE; B.g_active[g] = 0; B.g_tasks_left[g] = 0; }
}

// chunks returned by earlier launches become takeable (entry points outside the round call this first)
__global__ void k_pool_commit(SearchBuffers B) { pool_commit(B); }

__global__ void k_pending(SearchParams P, SearchBuffers B)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < P.G && B.g_phase[g] == PH_SEARCH) atomicAdd(B.pending, 1);
}

// queue rows that hold a new leaf (a position the network has to evaluate) after a round, compacted; rows[] order is
// arbitrary.  counts[0] = searches still running, counts[1] = rows written.
__global__ void k_leaf_rows(SearchParams P, SearchBuffers B, int32_t* __restrict__ rows, int32_t* __restrict__ counts)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= P.G * P.K) return;
    const int g = slot / P.K;
    if (B.g_phase[g] != PH_SEARCH) return;
    if (slot == g * P.K) atomicAdd(&counts[0], 1);
    if (B.s_state[slot] == SIM_LEAF) rows[atomicAdd(&counts[1], 1)] = slot;
}

// statistics of the root, or of the node reached from the root along path[g][0 .. path_len) (move labels, NOMOVE ends
// the path early); a node that is not linked in the tree reports count 0
__global__ __launch_bounds__(64) void k_root_stats(SearchParams P, SearchBuffers B, const uint16_t* __restrict__ path,
                                                  int path_len, uint16_t* __restrict__ moves,
                                                  int32_t* __restrict__ n, double* __restrict__ w,
                                                  float* __restrict__ p, int32_t* __restrict__ sum_n,
                                                  uint8_t* __restrict__ counts)
{
    __shared__ uint32_t chtab[MAX_CHUNKS];
    const int g = blockIdx.x;
    if (g >= P.G) return;
    const GameView gv = make_view(B, P, g, B.counters + (size_t)g * CT_COUNT, chtab);   // counts go straight to the global block
    const int lane = lane_id();
    int root = B.g_root[g];
    for (int d = 0; path && d < path_len && root >= 0; ++d) {
        const uint16_t mv = path[(size_t)g * path_len + d];
        if (mv == NOMOVE) break;
        char* base = rec_ptr(gv, (uint32_t)root);
        const NodeHdr hdr = load_hdr(base);
        const int pn = (int)(hdr.meta & 0xFF);
        const uint16_t* pm = node_mv(base, pn);
        int child = -1;
        if (hdr.stat) {
            const EdgeStat* sb = edge_ptr(gv, hdr.stat);
            for (int j = lane; j < pn; j += 64)
                if (pm[j] == mv) child = sb[j].child;
        }
        const unsigned long long hit = __ballot(child >= 0);
        root = hit ? __shfl(child, __ffsll((long long)hit) - 1) : -1;
    }
    int nm = 0, root_sum_n = 0;
    char* base = nullptr;
    const EdgeStat* sb = nullptr;
    if (root >= 0) {
        base = rec_ptr(gv, (uint32_t)root);
        const NodeHdr hdr = load_hdr(base);
        nm = (int)(hdr.meta & 0xFF);
        root_sum_n = hdr.sum_n;
        if (hdr.stat) sb = edge_ptr(gv, hdr.stat);
    }
    for (int j = lane; j < MAXMOVES; j += 64) {
        const bool ok = j < nm;
        EdgeStat es{0.0, 0, CHILD_UNKNOWN};
        if (ok && sb) es = sb[j];
        if (moves) moves[(size_t)g * MAXMOVES + j] = ok ? node_mv(base, nm)[j] : (uint16_t)NOMOVE;
        if (n) n[(size_t)g * MAXMOVES + j] = es.n;
        if (w) w[(size_t)g * MAXMOVES + j] = es.w;
        if (p) p[(size_t)g * MAXMOVES + j] = ok ? node_p(base)[j] : 0.0f;
    }
    if (lane == 0) {
        if (sum_n) sum_n[g] = root_sum_n;
        if (counts) counts[g] = (uint8_t)nm;
    }
}

// Principal variation (print_depth_info, player.py:408-433): from the root follow the most-visited edge -- `>=` keeps
// the LAST maximum, banned moves are skipped at the root only -- until a node that was never selected from (the
// reference creates a node's edges at its first selection: an empty `a` ends the line), a terminal / unlinked child or
// max_len moves.  moves [G][max_len] (NOMOVE padded), visits [G][max_len].
__global__ __launch_bounds__(64) void k_pv(SearchParams P, SearchBuffers B, int max_len, uint16_t* __restrict__ moves,
                                          int32_t* __restrict__ visits)
{
    __shared__ uint32_t chtab[MAX_CHUNKS];
    const int g = blockIdx.x;
    if (g >= P.G) return;
    const GameView gv = make_view(B, P, g, B.counters + (size_t)g * CT_COUNT, chtab);
    const int lane = lane_id();
    const int n_no_act = B.g_n_no_act[g];
    const uint16_t* no_act = B.g_no_act + (size_t)g * MAX_NO_ACT;
    int node = B.g_root[g];
    int d = 0;
    for (; d < max_len && node >= 0; ++d) {
        char* base = rec_ptr(gv, (uint32_t)node);
        const NodeHdr hdr = load_hdr(base);
        const int nm = (int)(hdr.meta & 0xFF);
        if (hdr.stat == 0u || nm == 0) break;
        const uint16_t* pm = node_mv(base, nm);
        const EdgeStat* sb = edge_ptr(gv, hdr.stat);
        int best_n = 0, best_j = -1;                          // per lane: last j with n >= running maximum
        for (int j = lane; j < nm; j += 64) {
            const int n = sb[j].n;
            bool banned = false;
            if (d == 0) for (int k = 0; k < n_no_act; ++k) banned = banned || (no_act[k] == pm[j]);
            if (!banned && n >= best_n) { best_n = n; best_j = j; }
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {                   // arg max of (n, j)
            const int on = __shfl_xor(best_n, s, 64), oj = __shfl_xor(best_j, s, 64);
            if (oj >= 0 && (best_j < 0 || on > best_n || (on == best_n && oj > best_j))) { best_n = on; best_j = oj; }
        }
        best_n = uni(best_n); best_j = uni(best_j);
        if (best_j < 0 || best_n <= 0) break;
        if (lane == 0) {
            moves[(size_t)g * max_len + d] = pm[best_j];
            visits[(size_t)g * max_len + d] = best_n;
        }
        node = uni(sb[best_j].child);
    }
    for (int i = d + lane; i < max_len; i += 64) {
        moves[(size_t)g * max_len + i] = (uint16_t)NOMOVE;
        visits[(size_t)g * max_len + i] = 0;
    }
}

__global__ __launch_bounds__(64) void k_choose(SearchParams P, SearchBuffers B, const double* __restrict__ u,
                                              int32_t* __restrict__ action)
{
    __shared__ SearchLDS L;
    const int g = blockIdx.x;
    if (g >= P.G) return;
    const GameView gv = make_view(B, P, g, B.counters + (size_t)g * CT_COUNT, L.chtab);   // counts go straight to the global block
    const int a = choose_action(P, B, gv, L, u ? u[g] : 0.5, B.g_enable_resign[g] != 0);
    if (lane_id() == 0) action[g] = a;
}

// cz_search_stop: no further simulations are STARTED; the ones in flight are backed up by the next round
__global__ void k_stop(SearchParams P, SearchBuffers B)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < P.G && B.g_phase[g] == PH_SEARCH) B.g_tasks_left[g] = 0;
}

// test hook: `n` draws of the root noise exactly as k_noise produces them (same generator, same stream addressing: one
// counter sequence per (epoch, sim, move) triple)
__global__ void k_debug_noise(uint64_t seed, uint32_t game_key, float alpha, int nm, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t epoch = (uint32_t)(i / (8 * 128)), sim = (uint32_t)(i / 128) % 8u, j = (uint32_t)i % 128u;
    NoiseRng rng = NoiseRng::make(seed, game_key, epoch, sim, j);
    out[i] = dirichlet0(alpha, nm, rng);
}

__global__ void k_debug_sqrt(const int32_t* __restrict__ x, double* __restrict__ y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __dsqrt_rn((double)(x[i] + 1));
}

}  // namespace

// ---- host side: the opaque search object ----------------------------------------------------------------------
struct cz_search {
    SearchParams P;
    SearchBuffers B;
    void* slab = nullptr;             // everything but the chunk pool
    void* pool = nullptr;             // n_chunks x 1 MiB
    size_t bytes = 0;                 // slab + pool
    int device = 0;
    int prev_compact = 0;             // the previous round built a compact queue: its results are indexed by compact row
    int keep_chunks_created = 0;      // P.keep_chunks as sized at creation (cz_search_set_sims never goes below it)
};

namespace {

int serr(int code, const char* what)
{
    czi_set_error(what);
    return code;
}
int serr_hip(const char* what, hipError_t e)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    czi_set_error(buf);
    return CZ_ERR_HIP;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T>
void carve(char*& cur, T*& ptr, size_t count, bool dry)
{
    if (!dry) ptr = reinterpret_cast<T*>(cur);
    cur += align_up(sizeof(T) * count);
}

size_t layout(cz_search* s, char* base, bool dry)
{
    const SearchParams& P = s->P;
    SearchBuffers& B = s->B;
    char* cur = base;
    const size_t G = (size_t)P.G, H = (size_t)P.hash_cap;
    const size_t K = (size_t)P.K, D = (size_t)P.max_depth, PL = (size_t)P.max_plies + 2;
    carve(cur, B.hash_tab, G * H, dry);                  // first: the only part that has to be zeroed per game
    carve(cur, B.pool_ring, (size_t)P.n_chunks, dry);
    carve(cur, B.pool_head, 1, dry);
    carve(cur, B.pool_tail, 1, dry);
    carve(cur, B.pool_tail_vis, 1, dry);
    carve(cur, B.g_chunk_tab, G * (size_t)P.max_chunks, dry);
    carve(cur, B.g_nchunks, G, dry);
    carve(cur, B.g_heap_top, G, dry);
    carve(cur, B.g_node_count, G, dry);
    carve(cur, B.g_root, G, dry);
    carve(cur, B.g_board, G * BOARD_LDS, dry);
    carve(cur, B.g_tasks_left, G, dry);
    carve(cur, B.g_active, G, dry);
    carve(cur, B.g_phase, G, dry);
    carve(cur, B.g_turns, G, dry);
    carve(cur, B.g_no_act, G * MAX_NO_ACT, dry);
    carve(cur, B.g_n_no_act, G, dry);
    carve(cur, B.g_increase_temp, G, dry);
    carve(cur, B.s_state, G * K, dry);
    carve(cur, B.s_depth, G * K, dry);
    carve(cur, B.s_node, G * K, dry);
    carve(cur, B.s_path_node, G * K * D, dry);
    carve(cur, B.s_path_edge, G * K * D, dry);
    carve(cur, B.g_game_id, G, dry);
    carve(cur, B.g_enable_resign, G, dry);
    carve(cur, B.g_no_eat, G, dry);
    carve(cur, B.g_hist_key, G * PL * KEY_WORDS, dry);
    carve(cur, B.g_hist_hash, G * PL, dry);
    carve(cur, B.g_hist_act, G * PL, dry);
    carve(cur, B.counters, G * CT_COUNT, dry);
    carve(cur, B.ring, (size_t)P.ring_cap * P.record_stride, dry);
    carve(cur, B.ring_tail, 1, dry);
    carve(cur, B.g_last_action, G, dry);
    carve(cur, B.pending, 1, dry);
    carve(cur, B.noise, G * K * MAXMOVES, dry);
    carve(cur, B.g_noise_epoch, G, dry);
    carve(cur, B.g_prev_board, G * BOARD_LDS, dry);
    carve(cur, B.g_hist_kind, G, dry);
    carve(cur, B.s_qrow, G * K, dry);
    return (size_t)(cur - base);
}

// chunks a game needs for one full search on an empty tree (what it keeps through resets and between games)
int keep_chunks_for(int sims)
{
    constexpr long long PER_TASK = NODE_HDR_GRANULES + (6 * RESERVE_MOVES + 15) / 16 + RESERVE_MOVES;
    constexpr long long USABLE = CHUNK_GRANULES - MAXMOVES;
    const long long need = (long long)(sims + 1) * PER_TASK + 1;
    return (int)((need + USABLE - 1) / USABLE);
}

// initial state of the pool: game g owns chunks [g * keep, (g + 1) * keep), the rest is free
int init_pool(cz_search* s)
{
    const SearchParams& P = s->P;
    const size_t G = (size_t)P.G;
    const int owned = P.G * P.keep_chunks;
    uint32_t* tab = new (std::nothrow) uint32_t[G * P.max_chunks]();
    uint32_t* ring = new (std::nothrow) uint32_t[(size_t)P.n_chunks]();
    int32_t* nch = new (std::nothrow) int32_t[G];
    uint32_t* top = new (std::nothrow) uint32_t[G];
    hipError_t e = hipSuccess;
    if (!tab || !ring || !nch || !top) e = hipErrorOutOfMemory;
    else {
        for (size_t g = 0; g < G; ++g) {
            for (int i = 0; i < P.keep_chunks; ++i) tab[g * P.max_chunks + i] = (uint32_t)(g * P.keep_chunks + i);
            nch[g] = P.keep_chunks;
            top[g] = 1u;
        }
        for (int i = owned; i < P.n_chunks; ++i) ring[i - owned] = (uint32_t)i;
        const unsigned int head = 0u, tail = (unsigned int)(P.n_chunks - owned);
        e = hipMemcpy(s->B.g_chunk_tab, tab, sizeof(uint32_t) * G * P.max_chunks, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->B.pool_ring, ring, sizeof(uint32_t) * (size_t)P.n_chunks, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->B.g_nchunks, nch, sizeof(int32_t) * G, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->B.g_heap_top, top, sizeof(uint32_t) * G, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->B.pool_head, &head, sizeof(head), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->B.pool_tail, &tail, sizeof(tail), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->B.pool_tail_vis, &tail, sizeof(tail), hipMemcpyHostToDevice);
    }
    delete[] tab; delete[] ring; delete[] nch; delete[] top;
    return e == hipSuccess ? CZ_OK : serr_hip("cz_search_create: pool initialisation", e);
}

}  // namespace

extern "C" {

int cz_search_create(const cz_search_cfg* c, cz_search** out)
{
    if (!c || !out) return serr(CZ_ERR_ARG, "cz_search_create: null argument");
    if (c->n_games < 1 || c->sims_per_round < 1 || c->simulation_num_per_move < 1)
        return serr(CZ_ERR_ARG, "cz_search_create: n_games, sims_per_round, simulation_num_per_move must be >= 1");
    cz_search* s = new (std::nothrow) cz_search();
    if (!s) return serr(CZ_ERR_NOMEM, "cz_search_create: host allocation failed");
    SearchParams& P = s->P;
    P.G = c->n_games;
    P.K = c->sims_per_round;
    P.sims = c->simulation_num_per_move;
    P.vl = c->virtual_loss;
    P.max_depth = c->max_depth > 0 ? c->max_depth : MAXD_LDS;      // simulations deeper than this are cut (counted)
    if (P.max_depth > MAXD_LDS) P.max_depth = MAXD_LDS;
    P.max_game_length = c->max_game_length > 0 ? c->max_game_length : 100;
    P.max_plies = 2 * P.max_game_length + 2;
    // the most nodes one game's tree is sized for: every simulation of every ply of the longest game expands one
    long long max_nodes = c->max_nodes_per_game > 0 ? c->max_nodes_per_game : (long long)(P.max_plies + 2) * P.sims;
    if (max_nodes < P.sims + 2) max_nodes = P.sims + 2;
    if (max_nodes > (1ll << 23)) max_nodes = 1ll << 23;
    int h = 128;
    while ((long long)h * 2 < max_nodes * 3) h <<= 1;              // load factor <= 2/3 at max_nodes
    P.hash_cap = h;
    P.keep_chunks = keep_chunks_for(P.sims);
    s->keep_chunks_created = P.keep_chunks;
    // lock-step batches one k_sim launch may START for a game: 1.  (A batch whose simulations all end on terminal or
    // repeated positions needs no evaluation and could be followed by the next at once, but then the whole launch
    // waits for the few waves that chain batches of the slowest kind of simulation: with 3, the sustained k_sim(SELECT)
    // launch was 0.78 ms mean / 2.3 ms p99 against a 0.64 ms median.)
    P.max_batches = 1;
    if (const char* e = getenv("CZ_MAX_BATCHES")) { const int v = atoi(e); if (v >= 1 && v <= 16) P.max_batches = v; }
    // chunk table: the whole-game tree at ~420 B per node (node record + its share of statistics blocks)
    const long long game_chunks = (max_nodes * 420 + (long long)CHUNK_BYTES - 1) / (long long)CHUNK_BYTES;
    long long mc = game_chunks + P.keep_chunks;
    if (mc < P.keep_chunks + 1) mc = P.keep_chunks + 1;
    if (mc > MAX_CHUNKS) mc = MAX_CHUNKS;
    if (P.keep_chunks > MAX_CHUNKS) { delete s; return serr(CZ_ERR_ARG, "cz_search_create: simulation_num_per_move too large for one game's chunk table"); }
    P.max_chunks = (int)mc;
    P.planes_dtype = c->planes_dtype;
    P.in_planes = c->use_history ? 28 : 14;
    P.mode = MODE_EXTERNAL;
    P.c_puct = c->c_puct;
    P.c_puct_f32 = (float)c->c_puct;
    P.noise_eps = c->noise_eps;
    P.one_minus_eps_f32 = (float)(1.0 - c->noise_eps);
    P.dirichlet_alpha = c->dirichlet_alpha;
    P.tau_decay_rate = c->tau_decay_rate;
    P.resign_threshold = c->resign_threshold;
    P.min_resign_turn = c->min_resign_turn;
    P.evaluate = c->evaluate;
    P.enable_resign_rate = c->enable_resign_rate;
    P.seed = c->seed;
    P.game_id_stride = (uint32_t)P.G;
    P.ring_cap = c->ring_capacity > 0 ? c->ring_capacity : 2 * P.G + 64;
    P.record_stride = (int)((sizeof(GameRecord) + sizeof(uint16_t) * (size_t)(P.max_plies + 2) + 15) & ~(size_t)15);
    if (P.planes_dtype < CZ_F32 || P.planes_dtype > CZ_U8) { delete s; return serr(CZ_ERR_ARG, "cz_search_create: planes_dtype"); }
    hipError_t e = hipGetDevice(&s->device);
    if (e != hipSuccess) { delete s; return serr_hip("cz_search_create: hipGetDevice", e); }
    // pool size: what the games can use at most, bounded by the memory that is free now (the caller's network and
    // queue buffers come later: leave a fifth of it), never less than what the games keep
    const long long floor_chunks = (long long)P.G * P.keep_chunks + 1;
    long long want = c->pool_chunks > 0 ? c->pool_chunks : (long long)P.G * P.max_chunks;
    if (c->pool_chunks <= 0) {
        size_t free_b = 0, total_b = 0;
        e = hipMemGetInfo(&free_b, &total_b);
        if (e != hipSuccess) { delete s; return serr_hip("cz_search_create: hipMemGetInfo", e); }
        P.n_chunks = 0;
        const size_t fixed = layout(s, nullptr, true);             // (n_chunks = 0: the ring is added below, 4 B per chunk)
        const long long avail = ((long long)(free_b / 5 * 4) - (long long)fixed) / (long long)(CHUNK_BYTES + 4);
        if (want > avail) want = avail;
    }
    if (want < floor_chunks) want = floor_chunks;
    if (want > 0x7fffff00ll) want = 0x7fffff00ll;
    P.n_chunks = (int)want;
    const size_t slab_bytes = layout(s, nullptr, true);
    e = hipMalloc(&s->slab, slab_bytes);
    if (e != hipSuccess) { delete s; return serr_hip("cz_search_create: hipMalloc", e); }
    e = hipMalloc(&s->pool, (size_t)P.n_chunks * CHUNK_BYTES + 4096);   // (+ slack: speculative reads past a record)
    if (e != hipSuccess) { (void)hipFree(s->slab); delete s; return serr_hip("cz_search_create: hipMalloc (chunk pool)", e); }
    s->bytes = slab_bytes + (size_t)P.n_chunks * CHUNK_BYTES;
    e = hipMemset(s->slab, 0, slab_bytes);
    if (e != hipSuccess) { (void)cz_search_destroy(s); return serr_hip("cz_search_create: hipMemset", e); }
    layout(s, (char*)s->slab, false);
    s->B.pool = (char*)s->pool;
    const int rc = init_pool(s);
    if (rc != CZ_OK) { (void)cz_search_destroy(s); return rc; }
    *out = s;
    return CZ_OK;
}

int cz_search_destroy(cz_search* s)
{
    if (!s) return CZ_OK;
    (void)hipFree(s->pool);
    (void)hipFree(s->slab);
    delete s;
    return CZ_OK;
}

size_t cz_search_bytes(const cz_search* s) { return s ? s->bytes : 0; }

int cz_search_info(const cz_search* s, int32_t* out)
{
    if (!s || !out) return serr(CZ_ERR_ARG, "cz_search_info: null argument");
    out[0] = s->P.G; out[1] = s->P.K; out[2] = s->P.sims; out[3] = s->P.n_chunks; out[4] = s->P.max_chunks;
    out[5] = s->P.hash_cap; out[6] = s->P.max_depth; out[7] = s->P.max_plies; out[8] = s->P.record_stride;
    out[9] = s->P.ring_cap; out[10] = CT_COUNT; out[11] = s->P.in_planes; out[12] = s->P.keep_chunks;
    out[13] = MAX_NO_ACT; out[14] = 0; out[15] = 0;
    return CZ_OK;
}

int cz_search_memory_info(cz_search* s, int64_t* host_out, void* stream)
{
    if (!s || !host_out) return serr(CZ_ERR_ARG, "cz_search_memory_info: null argument");
    const size_t G = (size_t)s->P.G;
    int32_t* nch = new (std::nothrow) int32_t[G];
    uint32_t* top = new (std::nothrow) uint32_t[G];
    int32_t* cnt = new (std::nothrow) int32_t[G];
    unsigned int ht[2] = {0u, 0u};
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = (nch && top && cnt) ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpyAsync(nch, s->B.g_nchunks, sizeof(int32_t) * G, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(top, s->B.g_heap_top, sizeof(uint32_t) * G, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(cnt, s->B.g_node_count, sizeof(int32_t) * G, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&ht[0], s->B.pool_head, sizeof(unsigned int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&ht[1], s->B.pool_tail, sizeof(unsigned int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) {
        int64_t held = 0, held_max = 0, used = 0, used_max = 0, nodes = 0, nodes_max = 0;
        for (size_t g = 0; g < G; ++g) {
            held += nch[g];
            if (nch[g] > held_max) held_max = nch[g];
            const int64_t u = (int64_t)(top[g] >> CHUNK_SHIFT) * (int64_t)CHUNK_BYTES + (int64_t)(top[g] & (CHUNK_GRANULES - 1)) * 16;
            used += u;
            if (u > used_max) used_max = u;
            nodes += cnt[g];
            if (cnt[g] > nodes_max) nodes_max = cnt[g];
        }
        host_out[0] = s->P.n_chunks;                  // pool size (chunks of 1 MiB)
        host_out[1] = (int64_t)(unsigned int)(ht[1] - ht[0]);     // free chunks
        host_out[2] = held;                           // chunks owned by games
        host_out[3] = held_max;                       // ... by the largest game
        host_out[4] = used;                           // tree bytes in use (all games)
        host_out[5] = used_max;                       // ... of the largest game
        host_out[6] = nodes;
        host_out[7] = nodes_max;
    }
    delete[] nch; delete[] top; delete[] cnt;
    if (e != hipSuccess) return serr_hip("cz_search_memory_info", e);
    return CZ_OK;
}

#define S_LAUNCH_CHECK(name)                                         \
    do {                                                             \
        hipError_t e_ = hipGetLastError();                           \
        if (e_ != hipSuccess) return serr_hip(name, e_);             \
    } while (0)

int cz_search_start_selfplay(cz_search* s, uint64_t seed, uint32_t first_game_id, uint32_t game_id_stride, void* stream)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_start_selfplay: null handle");
    s->P.mode = MODE_SELFPLAY;
    s->P.seed = seed;
    s->P.game_id_stride = game_id_stride ? game_id_stride : (uint32_t)s->P.G;
    hipError_t e = hipMemsetAsync(s->B.ring_tail, 0, sizeof(unsigned int), (hipStream_t)stream);
    if (e != hipSuccess) return serr_hip("cz_search_start_selfplay", e);
    hipLaunchKernelGGL(k_pool_commit, dim3(1), dim3(1), 0, (hipStream_t)stream, s->B);
    hipLaunchKernelGGL(k_start_selfplay, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B, first_game_id);
    S_LAUNCH_CHECK("cz_search_start_selfplay");
    return CZ_OK;
}

int cz_search_set_roots(cz_search* s, const int8_t* boards, const int32_t* turns, const uint16_t* no_act,
                        const uint8_t* n_no_act, const uint8_t* increase_temp, const uint8_t* enable_resign,
                        const uint8_t* select_mask, const int8_t* prev_boards, const uint8_t* hist_kind, void* stream)
{
    if (!s || !boards) return serr(CZ_ERR_ARG, "cz_search_set_roots: null argument");
    s->P.mode = MODE_EXTERNAL;
    hipLaunchKernelGGL(k_pool_commit, dim3(1), dim3(1), 0, (hipStream_t)stream, s->B);
    hipLaunchKernelGGL(k_set_roots, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B, boards, turns, no_act,
                       n_no_act, increase_temp, enable_resign, select_mask, prev_boards, hist_kind);
    S_LAUNCH_CHECK("cz_search_set_roots");
    return CZ_OK;
}

static int search_round_impl(cz_search* s, const float* policy, const float* value, void* planes, int32_t* q_rows,
                             int32_t* q_count, void* stream)
{
    const dim3 grid(s->P.G), block(64);
    hipStream_t st = (hipStream_t)stream;
    const bool noise = s->P.noise_eps != 0.0;
    const int compact = q_rows ? 1 : 0;
    // the rows consumed now were written after the PREVIOUS round: by compact row if that round built a compact queue
    const int consume_compact = s->prev_compact;
    s->prev_compact = compact;
    const dim3 nblock(256);
    const bool hist = s->P.in_planes == 28;
    if (hist) hipLaunchKernelGGL(k_sim<true>, grid, block, 0, st, s->P, s->B, policy, value, planes, SIM_BACKUP, consume_compact, q_rows, q_count);
    else hipLaunchKernelGGL(k_sim<false>, grid, block, 0, st, s->P, s->B, policy, value, planes, SIM_BACKUP, consume_compact, q_rows, q_count);
    hipLaunchKernelGGL(k_advance, grid, block, 0, st, s->P, s->B);
    if (noise) hipLaunchKernelGGL(k_noise, grid, nblock, 0, st, s->P, s->B);
    if (hist) hipLaunchKernelGGL(k_sim<true>, grid, block, 0, st, s->P, s->B, policy, value, planes, SIM_SELECT, compact, q_rows, q_count);
    else hipLaunchKernelGGL(k_sim<false>, grid, block, 0, st, s->P, s->B, policy, value, planes, SIM_SELECT, compact, q_rows, q_count);
    S_LAUNCH_CHECK("cz_search_round");
    return CZ_OK;
}

int cz_search_round(cz_search* s, const float* policy, const float* value, void* planes, void* stream)
{
    if (!s || !planes || !policy || !value) return serr(CZ_ERR_ARG, "cz_search_round: null argument");
    return search_round_impl(s, policy, value, planes, nullptr, nullptr, stream);
}

int cz_search_round_q(cz_search* s, const float* policy, const float* value, void* planes, int32_t* q_rows,
                      int32_t* q_count, void* stream)
{
    if (!s || !planes || !policy || !value || !q_rows || !q_count)
        return serr(CZ_ERR_ARG, "cz_search_round_q: null argument");
    return search_round_impl(s, policy, value, planes, q_rows, q_count, stream);
}

int cz_search_set_sims(cz_search* s, int simulation_num_per_move)
{
    if (!s || simulation_num_per_move < 1) return serr(CZ_ERR_ARG, "cz_search_set_sims: bad argument");
    // (a search longer than the chunks a game can own ends in counted overflow_sims, it is not refused here)
    s->P.sims = simulation_num_per_move;
    // a game that has to drop its tree keeps enough chunks for one full search of the NEW length (never fewer than it
    // was created with: the initial pool layout gave every game that many)
    int keep = keep_chunks_for(simulation_num_per_move);
    if (keep > s->P.max_chunks) keep = s->P.max_chunks;
    if (keep > s->keep_chunks_created) s->P.keep_chunks = keep;
    else s->P.keep_chunks = s->keep_chunks_created;
    return CZ_OK;
}

int cz_search_leaf_masks(cz_search* s, uint32_t* masks)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_leaf_masks: null handle");
    s->B.leaf_masks = masks;
    if (!masks) s->B.leaf_planes_off = 0;
    return CZ_OK;
}

int cz_search_leaf_planes(cz_search* s, int on)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_leaf_planes: null handle");
    if (!on && !s->B.leaf_masks) return serr(CZ_ERR_ARG, "cz_search_leaf_planes: the planes can only be switched off while cz_search_leaf_masks is set");
    s->B.leaf_planes_off = on ? 0 : 1;
    return CZ_OK;
}

int cz_search_policy_logits(cz_search* s, int on)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_policy_logits: null handle");
    s->P.policy_logits = on ? 1 : 0;
    return CZ_OK;
}

int cz_search_reset_trees(cz_search* s, void* stream)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_reset_trees: null handle");
    hipLaunchKernelGGL(k_pool_commit, dim3(1), dim3(1), 0, (hipStream_t)stream, s->B);
    hipLaunchKernelGGL(k_reset_trees, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B);
    S_LAUNCH_CHECK("cz_search_reset_trees");
    return CZ_OK;
}

int cz_search_pending(cz_search* s, int* host_out, void* stream)
{
    if (!s || !host_out) return serr(CZ_ERR_ARG, "cz_search_pending: null argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(s->B.pending, 0, sizeof(int32_t), st);
    if (e != hipSuccess) return serr_hip("cz_search_pending", e);
    hipLaunchKernelGGL(k_pending, dim3((s->P.G + 255) / 256), dim3(256), 0, st, s->P, s->B);
    S_LAUNCH_CHECK("cz_search_pending");
    int32_t v = 0;
    e = hipMemcpyAsync(&v, s->B.pending, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return serr_hip("cz_search_pending", e);
    *host_out = v;
    return CZ_OK;
}

int cz_search_leaf_rows(cz_search* s, int32_t* rows, int32_t* counts_dev, int* host_out, void* stream)
{
    if (!s || !rows || !counts_dev || !host_out) return serr(CZ_ERR_ARG, "cz_search_leaf_rows: null argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts_dev, 0, 2 * sizeof(int32_t), st);
    if (e != hipSuccess) return serr_hip("cz_search_leaf_rows", e);
    const int n = s->P.G * s->P.K;
    hipLaunchKernelGGL(k_leaf_rows, dim3((n + 255) / 256), dim3(256), 0, st, s->P, s->B, rows, counts_dev);
    S_LAUNCH_CHECK("cz_search_leaf_rows");
    int32_t v[2] = {0, 0};
    e = hipMemcpyAsync(v, counts_dev, sizeof(v), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return serr_hip("cz_search_leaf_rows", e);
    host_out[0] = v[0];
    host_out[1] = v[1];
    return CZ_OK;
}

int cz_search_root_stats(cz_search* s, uint16_t* moves, int32_t* n, double* w, float* p, int32_t* sum_n,
                         uint8_t* counts, void* stream)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_root_stats: null handle");
    hipLaunchKernelGGL(k_root_stats, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B, (const uint16_t*)nullptr, 0,
                       moves, n, w, p, sum_n, counts);
    S_LAUNCH_CHECK("cz_search_root_stats");
    return CZ_OK;
}

int cz_search_stop(cz_search* s, void* stream)
{
    if (!s) return serr(CZ_ERR_ARG, "cz_search_stop: null handle");
    hipLaunchKernelGGL(k_stop, dim3((s->P.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->P, s->B);
    S_LAUNCH_CHECK("cz_search_stop");
    return CZ_OK;
}

int cz_search_node_stats(cz_search* s, const uint16_t* path, int path_len, uint16_t* moves, int32_t* n, double* w,
                         float* p, int32_t* sum_n, uint8_t* counts, void* stream)
{
    if (!s || path_len < 0 || (path_len > 0 && !path)) return serr(CZ_ERR_ARG, "cz_search_node_stats: bad argument");
    hipLaunchKernelGGL(k_root_stats, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B, path, path_len, moves, n, w,
                       p, sum_n, counts);
    S_LAUNCH_CHECK("cz_search_node_stats");
    return CZ_OK;
}

int cz_search_pv(cz_search* s, int max_len, uint16_t* moves, int32_t* visits, void* stream)
{
    if (!s || !moves || !visits || max_len < 1) return serr(CZ_ERR_ARG, "cz_search_pv: bad argument");
    hipLaunchKernelGGL(k_pv, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B, max_len, moves, visits);
    S_LAUNCH_CHECK("cz_search_pv");
    return CZ_OK;
}

int cz_search_choose(cz_search* s, const double* u, int32_t* action, void* stream)
{
    if (!s || !action) return serr(CZ_ERR_ARG, "cz_search_choose: null argument");
    hipLaunchKernelGGL(k_choose, dim3(s->P.G), dim3(64), 0, (hipStream_t)stream, s->P, s->B, u, action);
    S_LAUNCH_CHECK("cz_search_choose");
    return CZ_OK;
}

int cz_search_counters(cz_search* s, uint64_t* host_out, void* stream)
{
    if (!s || !host_out) return serr(CZ_ERR_ARG, "cz_search_counters: null argument");
    const size_t n = (size_t)s->P.G * CT_COUNT;
    unsigned long long* tmp = new (std::nothrow) unsigned long long[n];
    if (!tmp) return serr(CZ_ERR_NOMEM, "cz_search_counters: host allocation failed");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(tmp, s->B.counters, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { delete[] tmp; return serr_hip("cz_search_counters", e); }
    for (int c = 0; c < CT_COUNT; ++c) host_out[c] = 0;
    for (int g = 0; g < s->P.G; ++g)
        for (int c = 0; c < CT_COUNT; ++c) {
            const unsigned long long v = tmp[(size_t)g * CT_COUNT + c];
            if (c == CT_MAX_DEPTH) { if (v > host_out[c]) host_out[c] = v; }
            else host_out[c] += v;
        }
    delete[] tmp;
    return CZ_OK;
}

int cz_search_game_counters(cz_search* s, uint64_t* host_out, void* stream)
{
    if (!s || !host_out) return serr(CZ_ERR_ARG, "cz_search_game_counters: null argument");
    static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "counters are 64-bit");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(host_out, s->B.counters, (size_t)s->P.G * CT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return serr_hip("cz_search_game_counters", e);
    return CZ_OK;
}

int cz_search_drain_records(cz_search* s, unsigned int* cursor, void* host_buf, int max_records, int* n_out, void* stream)
{
    if (!s || !cursor || !host_buf || !n_out) return serr(CZ_ERR_ARG, "cz_search_drain_records: null argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned int tail = 0;
    hipError_t e = hipMemcpyAsync(&tail, s->B.ring_tail, sizeof(tail), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return serr_hip("cz_search_drain_records", e);
    unsigned int cur = *cursor;
    if (tail - cur > (unsigned)s->P.ring_cap) cur = tail - (unsigned)s->P.ring_cap;   // overwritten records are lost
    int n = 0;
    while (cur != tail && n < max_records) {
        const size_t off = (size_t)(cur % (unsigned)s->P.ring_cap) * s->P.record_stride;
        e = hipMemcpyAsync((char*)host_buf + (size_t)n * s->P.record_stride, s->B.ring + off, s->P.record_stride,
                           hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return serr_hip("cz_search_drain_records", e);
        ++cur; ++n;
    }
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return serr_hip("cz_search_drain_records", e);
    *cursor = cur;
    *n_out = n;
    return CZ_OK;
}

int cz_debug_noise(uint64_t seed, uint32_t game_key, double alpha, int n_moves, double* out, int n, void* stream)
{
    if (n <= 0) return CZ_OK;
    if (!out || n_moves < 1 || n_moves > MAXMOVES || !(alpha > 0.0)) return serr(CZ_ERR_ARG, "cz_debug_noise: bad argument");
    hipLaunchKernelGGL(k_debug_noise, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, game_key, (float)alpha,
                       n_moves, out, n);
    S_LAUNCH_CHECK("cz_debug_noise");
    return CZ_OK;
}

int cz_debug_sqrt(const int32_t* x, double* y, int n, void* stream)
{
    if (n <= 0) return CZ_OK;
    hipLaunchKernelGGL(k_debug_sqrt, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, n);
    S_LAUNCH_CHECK("cz_debug_sqrt");
    return CZ_OK;
}

}  // extern "C"
