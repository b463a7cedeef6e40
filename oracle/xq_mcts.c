/*
 * oracle/xq_mcts.c -- TEST INFRASTRUCTURE ONLY (see xq_mcts.h).
 *
 * Restates cchess_alphazero/agent/player.py (VisitState/ActionState :17-33, action :145-196,
 * MCTS_search :198-260, select_action_q_and_u :262-320, update_tree :340-373, calc_policy
 * :375-406, apply_temperature :453-470) and worker/self_play.py:95-212 in plain C.
 *
 * Arithmetic types follow what CPython + NumPy 2 do in the reference:
 *   priors float32 (sum and quotient in float32, in legal-move order); sqrt in float64;
 *   non-root U term: float32(c_puct)*p in float32, then float64; root: float64 throughout;
 *   W/Q in float64; visit counts integers.  Compile with -ffp-contract=off.
 */
#include "xq_mcts.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define MAXDEPTH 1024

typedef struct Node {
    int8_t board[90];
    int sum_n;
    int n_moves;
    int waiting;
    int spread;            /* priors already pushed to the edges */
    uint16_t *moves;
    int32_t *n;
    double *w;
    float *p;
    float *pending;        /* NN policy row not yet spread (node.p) */
    int *parked;           /* indices of parked sims */
    int n_parked, cap_parked;
    struct Node *next;
} Node;

typedef struct {
    int active;            /* still has to back up */
    int depth;
    Node *path_node[MAXDEPTH];
    int path_edge[MAXDEPTH];
    int8_t board[90];      /* current state */
    Node *leaf;            /* expanded this round, waiting for NN */
    int fresh;             /* started by action() (not resumed from a parked state): player.py:217 */
    int deferred; double dval; /* ended on a terminal / repeated position: value waiting for its update_tree task */
} Sim;

struct xqo_player {
    xqo_play_cfg cfg;
    int enable_resign;
    xqo_eval_fn fn; void *fn_ctx;
    xqo_rng_fn rng; void *rng_ctx;
    uint64_t noise_ctr;
    Node **buckets; size_t nbuckets; int n_nodes; long n_edges;
    int8_t root[90];
    const uint16_t *no_act; int n_no_act;
    Sim *sims; int n_sims_cap;
    xqo_counters ctr;
    int hist_kind; int8_t hist_prev[90];
};

/* ---- tree ---------------------------------------------------------------- */
static uint64_t board_hash(const int8_t *b)
{
    uint64_t h = 1469598103934665603ULL;
    int i;
    for (i = 0; i < 90; i++) { h ^= (uint8_t)b[i]; h *= 1099511628211ULL; }
    return h;
}

static Node *tree_find(const xqo_player *p, const int8_t *b)
{
    Node *n = p->buckets[board_hash(b) & (p->nbuckets - 1)];
    while (n) { if (memcmp(n->board, b, 90) == 0) return n; n = n->next; }
    return 0;
}

static Node *tree_insert(xqo_player *p, const int8_t *b)
{
    Node *n = (Node *)calloc(1, sizeof(Node));
    uint16_t mv[XQO_MAXMOVES];
    size_t k = board_hash(b) & (p->nbuckets - 1);
    int c = xqo_legal_moves(b, mv);
    if (c > XQO_MAXMOVES) c = XQO_MAXMOVES;
    memcpy(n->board, b, 90);
    n->n_moves = c;
    n->moves = (uint16_t *)malloc(sizeof(uint16_t) * (c ? c : 1));
    memcpy(n->moves, mv, sizeof(uint16_t) * c);
    n->n = (int32_t *)calloc(c ? c : 1, sizeof(int32_t));
    n->w = (double *)calloc(c ? c : 1, sizeof(double));
    n->p = (float *)calloc(c ? c : 1, sizeof(float));
    n->next = p->buckets[k];
    p->buckets[k] = n;
    p->n_nodes++;
    p->n_edges += c;
    return n;
}

static void node_free(Node *n)
{
    free(n->moves); free(n->n); free(n->w); free(n->p); free(n->pending); free(n->parked); free(n);
}

xqo_player *xqo_player_create(const xqo_play_cfg *cfg, int enable_resign, xqo_eval_fn fn, void *fn_ctx,
                              xqo_rng_fn rng, void *rng_ctx)
{
    xqo_player *p = (xqo_player *)calloc(1, sizeof(*p));
    xqo_init();
    p->cfg = *cfg;
    p->enable_resign = enable_resign;
    p->fn = fn; p->fn_ctx = fn_ctx; p->rng = rng; p->rng_ctx = rng_ctx;
    p->nbuckets = 1u << 16;
    p->buckets = (Node **)calloc(p->nbuckets, sizeof(Node *));
    p->n_sims_cap = cfg->search_threads > 0 ? cfg->search_threads : 1;
    p->sims = (Sim *)calloc(p->n_sims_cap, sizeof(Sim));
    return p;
}

void xqo_player_destroy(xqo_player *p)
{
    size_t i;
    if (!p) return;
    for (i = 0; i < p->nbuckets; i++) {
        Node *n = p->buckets[i];
        while (n) { Node *nx = n->next; node_free(n); n = nx; }
    }
    free(p->buckets); free(p->sims); free(p);
}

void xqo_player_counters(const xqo_player *p, xqo_counters *out) { *out = p->ctr; }
void xqo_player_set_history(xqo_player *p, int kind, const int8_t prev[90])
{
    p->hist_kind = kind;
    if (kind == 1 && prev) memcpy(p->hist_prev, prev, 90);
}
int xqo_player_tree_size(const xqo_player *p) { return p->n_nodes; }

/* ---- Dirichlet(alpha * 1_n)[0] ~ Beta(alpha, alpha (n-1)) (player.py:304).  The reference draws
 * from NumPy's global RNG, which cannot be bit-matched; this is a distributional restatement. */
static double rng_u(xqo_player *p) { return p->rng ? p->rng(p->rng_ctx, 2, p->noise_ctr++) : 0.5; }

static double gamma_draw(xqo_player *p, double a)
{
    double boost = 1.0, d, c;
    if (a <= 0) return 0.0;
    if (a < 1.0) { double u = rng_u(p); if (u <= 0) u = 1e-300; boost = pow(u, 1.0 / a); a += 1.0; }
    d = a - 1.0 / 3.0; c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double u1 = rng_u(p), u2 = rng_u(p), x, v, u;
        if (u1 <= 0) u1 = 1e-300;
        x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);   /* Box-Muller */
        v = 1.0 + c * x;
        if (v <= 0) continue;
        v = v * v * v;
        u = rng_u(p);
        if (u < 1.0 - 0.0331 * x * x * x * x) return boost * d * v;
        if (u > 0 && log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
    }
}

static double dirichlet0(xqo_player *p, double alpha, int n)
{
    double x = gamma_draw(p, alpha), y = n > 1 ? gamma_draw(p, alpha * (n - 1)) : 0.0;
    return (x + y) > 0 ? x / (x + y) : 1.0 / n;
}

/* ---- select_action_q_and_u: player.py:262-320 ------------------------------ */
static int is_banned(const xqo_player *p, uint16_t mv)
{
    int i;
    for (i = 0; i < p->n_no_act; i++) if (p->no_act[i] == mv) return 1;
    return 0;
}

static void spread_priors(Node *node)
{
    /* :272-284 -- float32 accumulation in legal-move order, then float32 division */
    int i;
    float all_p = 0.0f;
    int first = 1;
    for (i = 0; i < node->n_moves; i++) {
        float mp = node->pending[node->moves[i]];
        node->p[i] = mp;
        if (first) { all_p = mp; first = 0; }      /* int 0 + float32 == float32(mp) */
        else all_p = all_p + mp;
    }
    if (all_p == 0.0f) all_p = 1.0f;
    for (i = 0; i < node->n_moves; i++) node->p[i] = node->p[i] / all_p;
    free(node->pending);
    node->pending = 0;
    node->spread = 1;
}

static int select_edge(xqo_player *p, Node *node)
{
    const int is_root = memcmp(node->board, p->root, 90) == 0;        /* :266 */
    const double xx = sqrt((double)(node->sum_n + 1));                 /* np.sqrt(int) -> float64 */
    const double e = p->cfg.noise_eps, c_puct = p->cfg.c_puct;
    double best_score = -99999999.0;
    int best = -1, i;
    if (node->pending) spread_priors(node);
    for (i = 0; i < node->n_moves; i++) {
        double q, score, u;
        if (is_root && p->n_no_act && is_banned(p, node->moves[i])) continue;      /* :298-300 */
        q = node->n[i] ? node->w[i] / (double)node->n[i] : 0.0;
        if (is_root) {
            /* (1 - e) * p_ : python float * np.float32 -> float32; + e * float64 -> float64 (:304) */
            float a = (float)(1.0 - e) * node->p[i];
            double p_ = (double)a + e * dirichlet0(p, p->cfg.dirichlet_alpha, node->n_moves);
            u = c_puct * p_ * xx / (double)(1 + node->n[i]);
        } else {
            float a = (float)c_puct * node->p[i];                                   /* float32 product */
            u = (double)a * xx / (double)(1 + node->n[i]);
        }
        score = q + u;
        if (q > (1.0 - 1e-7)) { best = i; break; }                                  /* :309-311 */
        if (score >= best_score) { best_score = score; best = i; }                  /* :312-314 */
    }
    return best;
}

/* ---- update_tree: player.py:340-373 ---------------------------------------- */
static void backup(xqo_player *p, Sim *s, double v)
{
    const int vl = p->cfg.virtual_loss;
    int i;
    for (i = s->depth - 1; i >= 0; i--) {
        Node *node = s->path_node[i];
        int e = s->path_edge[i];
        v = -v;
        node->n[e] += 1 - vl;
        node->w[e] = node->w[e] + (v + (double)vl);
    }
    p->ctr.sims++;
    p->ctr.sum_depth += (uint64_t)s->depth;
    if ((uint64_t)s->depth > p->ctr.max_depth) p->ctr.max_depth = (uint64_t)s->depth;
    s->active = 0;
}

/* ---- MCTS_search: player.py:198-260 (one descent until the sim stops) -------- */
static void park(Node *node, int idx)
{
    if (node->n_parked == node->cap_parked) {
        node->cap_parked = node->cap_parked ? node->cap_parked * 2 : 4;
        node->parked = (int *)realloc(node->parked, sizeof(int) * node->cap_parked);
    }
    node->parked[node->n_parked++] = idx;
}

/* A simulation that ends on a terminal or repeated position does not update the tree itself: MCTS_search SUBMITS
 * update_tree to the executor (player.py:206, :228-232) and returns.  That task sits in the executor's queue behind the
 * search tasks of the batch, which were all submitted first (player.py:173-174), and the interpreter lets a search
 * thread run its descent to the end before it switches: in the unmodified reference the K descents of a batch see each
 * other's virtual losses but none of the batch's terminal results.  Canonical order: such a value
 * is applied when the descents of the phase are over, in index order -- tests/golden/kgt1_spread.json: with this rule
 * the canonical result IS one of the visit vectors the reference itself produces in 23 of 24 recorded cases. */
static void finish(Sim *s, double v)
{
    s->deferred = 1;
    s->dval = v;
}
static void flush_deferred(xqo_player *p, int n)
{
    int i;
    for (i = 0; i < n; i++)
        if (p->sims[i].deferred) { p->sims[i].deferred = 0; backup(p, &p->sims[i], p->sims[i].dval); }
}
static void descend(xqo_player *p, int idx)
{
    Sim *s = &p->sims[idx];
    const int vl = p->cfg.virtual_loss;
    for (;;) {
        int over, v, fm, ck, i, e;
        Node *node;
        xqo_done(s->board, 0, &over, &v, &fm, &ck);
        if (over) {                                                   /* :204-208, value doubled */
            p->ctr.terminal_sims++;
            finish(s, (double)(v * 2));
            return;
        }
        node = tree_find(p, s->board);
        if (!node) {                                                  /* :211-221 expand */
            node = tree_insert(p, s->board);
            node->sum_n = 1;
            node->waiting = 1;
            s->leaf = node;
            p->ctr.expansions++;
            p->ctr.sum_leaf_moves += (uint64_t)node->n_moves;
            return;
        }
        for (i = 0; i < s->depth; i++)                                /* :223-236 state in history[:-1] */
            if (s->path_node[i] == node) break;
        if (i < s->depth) {
            int mv = node->moves[s->path_edge[i]];
            double val;
            if (xqo_will_check_or_catch(s->board, mv)) val = -1;
            else if (xqo_be_catched(s->board, mv)) val = 1;
            else val = 0;
            p->ctr.repetition_sims++;
            finish(s, val);
            return;
        }
        if (node->waiting) {                                          /* :238-242 */
            s->fresh = 0;                                             /* resumed by MCTS_search(state, hist) */
            park(node, idx);
            p->ctr.parked++;
            return;
        }
        e = select_edge(p, node);                                     /* :243 */
        if (e < 0) { backup(p, s, 0.0); return; }                     /* best_action None: cannot happen */
        node->sum_n += 1;                                             /* :245-252 */
        node->n[e] += vl;
        node->w[e] = node->w[e] - (double)vl;
        p->ctr.sum_edges_visited += (uint64_t)node->n_moves;
        if (s->depth >= MAXDEPTH) { backup(p, s, 0.0); return; }
        s->path_node[s->depth] = node;
        s->path_edge[s->depth] = e;
        s->depth++;
        {
            int8_t nb[90];
            xqo_step(s->board, node->moves[e], nb, 0);
            memcpy(s->board, nb, 90);
        }
    }
}

static void run_batch(xqo_player *p, int n)
{
    int i, pending;
    const int plane_len = p->cfg.use_history ? 28 * 90 : 14 * 90;
    float *planes = (float *)malloc(sizeof(float) * plane_len * (size_t)n);
    const int8_t **prevs = (const int8_t **)calloc((size_t)n, sizeof(*prevs));
    float *policy = (float *)malloc(sizeof(float) * XQO_NLABELS * (size_t)n);
    float *value = (float *)malloc(sizeof(float) * (size_t)n);
    int *leaf_sim = (int *)malloc(sizeof(int) * (size_t)n);
    for (i = 0; i < n; i++) {
        Sim *s = &p->sims[i];
        s->active = 1; s->depth = 0; s->leaf = 0; s->fresh = 1; s->deferred = 0;
        memcpy(s->board, p->root, 90);
    }
    for (i = 0; i < n; i++) descend(p, i);
    flush_deferred(p, n);
    for (;;) {
        int nl = 0, k;
        for (i = 0; i < n; i++) if (p->sims[i].leaf) leaf_sim[nl++] = i;
        if (!nl) break;
        for (k = 0; k < nl; k++) {
            Sim *s = &p->sims[leaf_sim[k]];
            if (!p->cfg.use_history) { xqo_planes(s->leaf->board, planes + (size_t)k * plane_len); continue; }
            /* expand_and_evaluate, player.py:322-338 */
            {
                const int8_t *prev = 0;
                if (s->fresh && p->hist_kind) prev = p->hist_kind == 1 ? p->hist_prev : 0;   /* real_hist[-5] */
                else if (s->depth >= 2) prev = s->path_node[s->depth - 2]->board;            /* history[-5] */
                xqo_planes_hist(s->leaf->board, prev, planes + (size_t)k * plane_len);
            }
        }
        p->fn(p->fn_ctx, planes, nl, plane_len, policy, value);
        p->ctr.nn_batches++;
        p->ctr.nn_positions += (uint64_t)nl;
        /* attach + backup in index order; collect parked sims */
        {
            int *resume = 0, nres = 0, cap = 0;
            for (k = 0; k < nl; k++) {
                Sim *s = &p->sims[leaf_sim[k]];
                Node *node = s->leaf;
                int j;
                node->pending = (float *)malloc(sizeof(float) * XQO_NLABELS);
                memcpy(node->pending, policy + (size_t)k * XQO_NLABELS, sizeof(float) * XQO_NLABELS);
                node->waiting = 0;
                for (j = 0; j < node->n_parked; j++) {
                    if (nres == cap) { cap = cap ? cap * 2 : 8; resume = (int *)realloc(resume, sizeof(int) * cap); }
                    resume[nres++] = node->parked[j];
                }
                node->n_parked = 0;
                s->leaf = 0;
                backup(p, s, (double)value[k]);          /* float(v) of a float32 */
            }
            /* resume parked sims in index order */
            {
                int a, b;
                for (a = 1; a < nres; a++) {
                    int t = resume[a];
                    for (b = a - 1; b >= 0 && resume[b] > t; b--) resume[b + 1] = resume[b];
                    resume[b + 1] = t;
                }
                for (a = 0; a < nres; a++) descend(p, resume[a]);
                flush_deferred(p, n);
            }
            free(resume);
        }
    }
    pending = 0;
    for (i = 0; i < n; i++) pending += p->sims[i].active;
    if (pending) { fprintf(stderr, "xq_mcts: %d sims never completed\n", pending); abort(); }
    free(planes); free(policy); free(value); free(leaf_sim); free(prevs);
}

/* ---- the one engine behaviour that is not in the reference: a game whose tree no longer fits the engine's chunk
 * pool starts over with an empty tree (counter tree_resets there).  Tests replay it by calling this at the plies
 * where the engine reported a reset. ---- */
void xqo_player_clear_tree(xqo_player *p)
{
    size_t i;
    for (i = 0; i < p->nbuckets; i++) {
        Node *n = p->buckets[i];
        while (n) { Node *nx = n->next; node_free(n); n = nx; }
        p->buckets[i] = 0;
    }
    p->n_nodes = 0;
    p->n_edges = 0;
}

/* ---- calc_policy: player.py:375-406 ------------------------------------------ */
static int calc_policy(xqo_player *p, const int8_t *board, int turns, double *policy)
{
    Node *node = tree_find(p, board);
    double max_q = -100.0, sum = 0.0;
    int i;
    memset(policy, 0, sizeof(double) * XQO_NLABELS);
    if (!node) return 0;
    if (node->spread || node->sum_n > 1) {
        for (i = 0; i < node->n_moves; i++) {
            double q;
            policy[node->moves[i]] = (double)node->n[i];
            if (p->n_no_act && is_banned(p, node->moves[i])) { policy[node->moves[i]] = 0.0; continue; }
            q = node->n[i] ? node->w[i] / (double)node->n[i] : 0.0;
            if (q > max_q) max_q = q;
        }
    }
    if (max_q < p->cfg.resign_threshold && p->enable_resign && turns > p->cfg.min_resign_turn)
        return 1;                                          /* resign: policy stays un-normalised */
    for (i = 0; i < XQO_NLABELS; i++) sum += policy[i];
    for (i = 0; i < XQO_NLABELS; i++) policy[i] /= sum;
    return 0;
}

int xqo_player_search(xqo_player *p, const int8_t board[90], int turns, const uint16_t *no_act, int n_no_act,
                      int increase_temp, double *policy)
{
    const int sims = p->cfg.simulation_num_per_move, K = p->cfg.search_threads > 0 ? p->cfg.search_threads : 1;
    Node *root;
    int done_n = 0, num_task;
    memcpy(p->root, board, 90);
    p->no_act = no_act; p->n_no_act = n_no_act;
    root = tree_find(p, board);
    if (root) done_n = root->sum_n;                                       /* :153-155 */
    if (n_no_act > 0 || increase_temp || done_n == sims) done_n = 0;      /* :156-158 */
    num_task = sims - done_n;
    if (num_task < 0) num_task = 0;
    if (num_task > 0) {
        int all = num_task, batch = all / K + (all % K != 0), it;
        for (it = 0; it < batch; it++) {
            int n = all - K * it;
            if (n > K) n = K;
            run_batch(p, n);
        }
    }
    return calc_policy(p, board, turns, policy);
}

/* apply_temperature + np.random.choice: player.py:453-470, :195 */
int xqo_sample_action(const xqo_play_cfg *cfg, const double *policy, int turns, int increase_temp, double u)
{
    double tau;
    int i;
    if (turns < 30 && cfg->tau_decay_rate != 0) tau = pow(cfg->tau_decay_rate, (double)(turns + 1));
    else tau = 0;
    if (tau < 0.1 || (turns >= 4 && cfg->evaluate)) tau = 0;
    if (increase_temp && !cfg->evaluate) tau = 0.5;
    if (tau == 0) {
        int best = 0;                                   /* np.argmax: first maximum in label order */
        for (i = 1; i < XQO_NLABELS; i++) if (policy[i] > policy[best]) best = i;
        return best;                                    /* choice over a one-hot vector */
    } else {
        static double ret[XQO_NLABELS];
        double sum = 0.0, c = 0.0, total;
        for (i = 0; i < XQO_NLABELS; i++) { ret[i] = policy[i] > 0 ? pow(policy[i], 1.0 / tau) : 0.0; sum += ret[i]; }
        for (i = 0; i < XQO_NLABELS; i++) ret[i] /= sum;
        total = 0.0;
        for (i = 0; i < XQO_NLABELS; i++) total += ret[i];      /* cdf[-1] */
        /* cdf = cumsum(p) / cdf[-1]; searchsorted(u, side='right'): first i with cdf[i] > u */
        for (i = 0; i < XQO_NLABELS; i++) {
            c += ret[i];
            if (c / total > u) return i;
        }
        for (i = XQO_NLABELS - 1; i >= 0; i--) if (ret[i] > 0) return i;
        return 0;
    }
}

int xqo_player_action(xqo_player *p, const int8_t board[90], int turns, const uint16_t *no_act, int n_no_act,
                      int increase_temp, double u, double *policy)
{
    int i;
    if (xqo_player_search(p, board, turns, no_act, n_no_act, increase_temp, policy)) return -1;
    for (i = 0; i < n_no_act; i++) policy[no_act[i]] = 0.0;
    return xqo_sample_action(&p->cfg, policy, turns, increase_temp, u);
}

int xqo_player_node_stats(const xqo_player *p, const int8_t board[90], uint16_t *moves, int32_t *n, double *w,
                          float *prior, int *sum_n)
{
    Node *node = tree_find(p, board);
    int i;
    if (!node) return -1;
    if (node->pending) spread_priors(node);
    for (i = 0; i < node->n_moves; i++) {
        if (moves) moves[i] = node->moves[i];
        if (n) n[i] = node->n[i];
        if (w) w[i] = node->w[i];
        if (prior) prior[i] = node->p[i];
    }
    if (sum_n) *sum_n = node->sum_n;
    return node->n_moves;
}

/* ---- crc32 (zlib polynomial) for per-ply visit fingerprints ------------------- */
static uint32_t crc32_update(uint32_t crc, const void *buf, size_t len)
{
    static uint32_t table[256];
    static int have = 0;
    const uint8_t *p = (const uint8_t *)buf;
    size_t i;
    if (!have) {
        uint32_t c; int n, k;
        for (n = 0; n < 256; n++) { c = (uint32_t)n; for (k = 0; k < 8; k++) c = c & 1 ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[n] = c; }
        have = 1;
    }
    crc ^= 0xFFFFFFFFu;
    for (i = 0; i < len; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFu;
}

/* ---- SelfPlayWorker.start_game: self_play.py:95-212 ---------------------------- */
int xqo_selfplay_game(const xqo_play_cfg *cfg, xqo_eval_fn fn, void *fn_ctx, xqo_rng_fn rng, void *rng_ctx,
                      uint16_t *moves_out, int max_plies, double *value_out, int *store_out,
                      xqo_counters *counters_out, uint32_t *visit_crc_out)
{
    const int enable_resign = rng(rng_ctx, 0, 0) > cfg->enable_resign_rate;       /* :102-105 */
    xqo_player *pl = xqo_player_create(cfg, enable_resign, fn, fn_ctx, rng, rng_ctx);
    int cap = max_plies + 4;
    int8_t (*hist)[90] = (int8_t (*)[90])malloc((size_t)(cap + 1) * 90);
    uint16_t *acts = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(cap + 1));
    double *policy = (double *)malloc(sizeof(double) * XQO_NLABELS);
    int8_t state[90];
    double value = 0;
    int turns = 0, game_over = 0, final_move = XQO_NOMOVE, no_eat_count = 0, check = 0, increase_temp = 0;
    uint16_t no_act[XQO_MAXMOVES];
    int n_no_act = 0, nstates = 1, store, i;

    xqo_state_to_board("rkemsmekr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RKEMSMEKR", state);
    memcpy(hist[0], state, 90);
    while (!game_over) {
        int8_t next[90];
        int no_eat, action, over, v;
        if (turns >= max_plies) { value = 0; break; }            /* guard only; never hit with sane configs */
        action = xqo_player_action(pl, state, turns, no_act, n_no_act, increase_temp,
                                   rng(rng_ctx, 1, (uint64_t)turns), policy);
        if (visit_crc_out) {
            uint16_t mv[XQO_MAXMOVES]; int32_t nn[XQO_MAXMOVES]; int sn = 0;
            int c = xqo_player_node_stats(pl, state, mv, nn, 0, 0, &sn);
            uint32_t crc = 0;
            if (c > 0) { crc = crc32_update(crc, mv, sizeof(uint16_t) * (size_t)c); crc = crc32_update(crc, nn, sizeof(int32_t) * (size_t)c); }
            visit_crc_out[turns] = crc;
        }
        if (action < 0) { value = -1; break; }                   /* resign :126-129 */
        acts[turns] = (uint16_t)action;
        if (xqo_step(state, action, next, &no_eat) != 0) { game_over = 1; value = 0; break; }   /* :135-141 */
        memcpy(state, next, 90);
        turns += 1;
        if (no_eat) no_eat_count += 1; else no_eat_count = 0;
        memcpy(hist[nstates++], state, 90);
        if (no_eat_count >= 120 || turns >= 2 * cfg->max_game_length) {    /* turns / 2 >= max_game_length */
            game_over = 1; value = 0;
        } else {
            int fm;
            xqo_done(state, 1, &over, &v, &fm, &check);
            game_over = over; value = v; final_move = fm;
            if (!game_over && !xqo_has_attack_chessman(state)) { game_over = 1; value = 0; }
            increase_temp = 0; n_no_act = 0;
            if (!game_over && !check) {
                int free_move = 0;
                for (i = 0; i < nstates - 1; i++) {              /* state in history[:-1] */
                    if (memcmp(hist[i], state, 90) != 0) continue;
                    if (xqo_will_check_or_catch(state, acts[i])) {
                        no_act[n_no_act++] = acts[i];
                    } else if (!xqo_be_catched(state, acts[i])) {
                        increase_temp = 1;
                        free_move += 1;
                        if (free_move >= 3) { game_over = 1; value = 0; break; }
                    }
                }
            }
        }
    }
    if (final_move != XQO_NOMOVE) {                              /* :177-184 */
        int8_t next[90];
        acts[turns] = (uint16_t)final_move;
        xqo_step(state, final_move, next, 0);
        memcpy(state, next, 90);
        turns += 1;
        value = -value;
    }
    if (turns % 2 == 1) value = -value;                          /* :190-191 */
    if (turns < 10) store = rng(rng_ctx, 0, 1) > 0.9; else store = 1;    /* :194-200 */
    for (i = 0; i < turns && i < max_plies + 1; i++) moves_out[i] = acts[i];
    *value_out = value;
    *store_out = store;
    if (counters_out) xqo_player_counters(pl, counters_out);
    xqo_player_destroy(pl);
    free(hist); free(acts); free(policy);
    return turns;
}

/* ---- stub networks and the shared counter-based RNG ----------------------------- */
static uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

void xqo_stub_uniform(void *ctx, const float *planes, int n, int plane_len, float *policy, float *value)
{
    (void)plane_len;
    const float v = ctx ? *(const float *)ctx : 0.0f;
    const float pu = (float)(1.0 / 2086.0);
    int i, a;
    (void)planes;
    for (i = 0; i < n; i++) {
        for (a = 0; a < XQO_NLABELS; a++) policy[(size_t)i * XQO_NLABELS + a] = pu;
        value[i] = v;
    }
}

void xqo_stub_hash(void *ctx, const float *planes, int n, int plane_len, float *policy, float *value)
{
    const uint64_t salt = ctx ? *(const uint64_t *)ctx : 0;
    int i, o, a;
    for (i = 0; i < n; i++) {
        const float *pl = planes + (size_t)i * plane_len;
        uint64_t h = salt;
        uint64_t uv;
        for (o = 0; o < plane_len; o++) if (pl[o] != 0.0f) h += mix64((uint64_t)o + 1);
        h = mix64(h);
        for (a = 0; a < XQO_NLABELS; a++) {
            uint64_t u = (mix64(h + (uint64_t)(a + 1) * 0x9E3779B97F4A7C15ULL) >> 40) & 0xFFFF;
            float x = (float)(u + 1) / 65536.0f;
            x = x * x; x = x * x; x = x * x;              /* x^8: a peaky, exactly reproducible prior */
            policy[(size_t)i * XQO_NLABELS + a] = x;
        }
        uv = (mix64(h ^ 0x5851F42D4C957F2DULL) >> 40) & 0xFFFF;
        value[i] = ((float)uv - 32768.0f) / 32768.0f;
    }
}

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    int r;
    for (r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

double xqo_philox_uniform(void *ctx, int stream, uint64_t idx)
{
    const uint64_t *sg = (const uint64_t *)ctx;      /* {seed, game_id} */
    uint32_t c[4] = { (uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream, (uint32_t)sg[1] };
    philox4x32_10(c, (uint32_t)sg[0], (uint32_t)(sg[0] >> 32));
    return ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) / 9007199254740992.0;
}
