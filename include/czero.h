/*
 * include/czero.h -- C-ABI of the MI355X-native Xiangqi self-play engine (libczero.so).
 *
 * The reference (NeymarL/ChineseChess-AlphaZero) is 100 % Python and has no FFI; this is the
 * seam a maintainer binds directly under its Python modules (ctypes stub: INTEGRATION.md).
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference's cchess_alphazero/ package).
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is CALLER-OWNED DEVICE memory (e.g. a torch
 *     tensor's data_ptr()) unless the parameter is documented as host memory;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls enqueue work and return;
 *   - return 0 (CZ_OK) or a negative CZ_ERR_* code, never throw; cz_last_error() is thread-local;
 *   - square s = y*9 + x (x 0..8, y 0..9, y = 0 is the side-to-move's back rank, as in
 *     environment/static_env.py:117-135); board = int8[90], 0 empty, +t mover / -t opponent,
 *     t = 1 pawn 2 cannon 3 rook 4 knight 5 elephant 6 advisor 7 king (Fen_2_Idx order + 1,
 *     environment/lookup_tables.py:27-42);
 *   - move = uint16 index into ActionLabelsRed (environment/lookup_tables.py:62-134), 0..2085;
 *     0xFFFF = none.
 */
#ifndef CZERO_H
#define CZERO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CZ_VERSION 2

#define CZ_OK 0
#define CZ_ERR_ARG (-1)
#define CZ_ERR_HIP (-2)
#define CZ_ERR_STATE (-3)
#define CZ_ERR_NOMEM (-4)

#define CZ_NSQ 90
#define CZ_NLABELS 2086
#define CZ_MAXMOVES 128
#define CZ_NOMOVE 0xFFFF

/* element type of the network-input planes written by the engine */
#define CZ_F32 0
#define CZ_F16 1
#define CZ_BF16 2
#define CZ_U8 3
#define CZ_F16C8 4   /* fp16 operand + c8 correction image (cz_conv3x3_c8): the residual-block entry points only */
#define CZ_F16C6 5   /* fp16 operand + c6 correction image (bf6 pieces; cz_conv3x3_c6_pack_weights): cz_resblock(_heads),
                        cz_input_resblock with 128 filters only */
#define CZ_F16C86 6  /* cz_resblock, 192 filters: the first c6 block of a tower whose input layer wrote a c8 image (x = c8 pair; first
                      * filter cz_conv3x3_c8_pack_weights', second cz_conv3x3_c6_pack_weights'; y = a c6 pair) */

int cz_version(void);
const char* cz_last_error(void);
int cz_device_count(void);

/* HOST buffers. label_of[90*90] (from*90+to -> label, 0xFFFF none), lab_ft[2086] (from<<8|to).
 * Replaces create_action_labels / ActionLabelsRed, environment/lookup_tables.py:62-134. */
int cz_label_tables(uint16_t* label_of, uint16_t* lab_ft);

/* ---- batched rules: one wavefront per board ------------------------------------------- */

/* get_legal_moves, environment/static_env.py:256-321 (pseudo-legal, reference emission order).
 * moves[n][128] (0xFFFF padded), counts[n]. */
int cz_movegen(const int8_t* boards, int n, uint16_t* moves, uint8_t* counts, void* stream);

/* done, environment/static_env.py:14-77.  over/v/final_move per board; check only when need_check
 * (may be NULL otherwise).  v is from the side to move's view. */
int cz_done(const int8_t* boards, int n, int need_check, int8_t* over, int8_t* v, uint16_t* final_move,
            uint8_t* check, void* stream);

/* step / new_step, environment/static_env.py:79-98: out = board after the move, flipped to the next
 * mover.  no_eat[i] = 1 no capture, 0 capture, 0xFF = the reference would raise ValueError (empty
 * source square or bad label; out = input board).  no_eat may be NULL. */
int cz_step(const int8_t* boards, const uint16_t* moves, int n, int8_t* out, uint8_t* no_eat, void* stream);

/* state_to_planes, environment/static_env.py:137-156: planes[n][14][10][9] of `dtype` (CZ_F32...). */
int cz_encode(const int8_t* boards, int n, void* planes, int dtype, void* stream);

/* will_check_or_catch, environment/static_env.py:390-421.  out[i] = 0/1, 0xFF = ValueError. */
int cz_check_or_catch(const int8_t* boards, const uint16_t* moves, int n, uint8_t* out, void* stream);

/* be_catched, environment/static_env.py:456-469. */
int cz_be_catched(const int8_t* boards, const uint16_t* moves, int n, uint8_t* out, void* stream);

/* has_attack_chessman, environment/static_env.py:471-479. */
int cz_has_attack(const int8_t* boards, int n, uint8_t* out, void* stream);

/* move-gen + done(need_check=True) + planes in one pass (the SURVEY 8(d) micro-suite kernel). */
int cz_rules_fused(const int8_t* boards, int n, uint16_t* moves, uint8_t* counts, int8_t* over, int8_t* v,
                   uint16_t* final_move, uint8_t* check, void* planes, int dtype, void* stream);

/* ---- batched PUCT-MCTS + self-play game loop: one wavefront per game ---------------------------
 * Replaces agent/player.py::CChessPlayer (action :145-196, MCTS_search :198-260,
 * select_action_q_and_u :262-320, expand_and_evaluate :322-338, update_tree :340-373, calc_policy
 * :375-406, apply_temperature :453-470) and worker/self_play.py::SelfPlayWorker.start_game :95-212
 * for n_games concurrent games.  The network stays with the caller: each cz_search_round() consumes the
 * policy/value rows of the previous round and writes the input planes of the new leaves; the evaluation
 * queue has one fixed slot per (game, simulation): slot = game * sims_per_round + sim.
 */
typedef struct cz_search cz_search;

typedef struct cz_search_cfg {
    int32_t n_games;                 /* G: concurrent games = wavefronts */
    int32_t sims_per_round;          /* K: config.play.search_threads (lock-step batch per game) */
    int32_t simulation_num_per_move; /* config.play.simulation_num_per_move */
    int32_t virtual_loss;            /* config.play.virtual_loss */
    int32_t max_nodes_per_game;      /* sizes a game's hash table and chunk table; 0 = (2 * max_game_length + 4) * sims,
                                        i.e. the tree of the longest game is kept whole (self_play.py:84,98-100) */
    int32_t pool_chunks;             /* tree memory shared by all games, in chunks of 1 MiB; 0 = what the games can use,
                                        at most 80 % of the device memory that is free at creation */
    int32_t max_depth;               /* longest path of one simulation; 0 = 64, at most 128 */
    int32_t max_game_length;         /* config.play.max_game_length (full moves) */
    int32_t planes_dtype;            /* CZ_F32 / CZ_F16 / CZ_BF16 / CZ_U8 */
    int32_t min_resign_turn;         /* config.play.min_resign_turn */
    int32_t evaluate;                /* config.opts.evaluate */
    int32_t ring_capacity;           /* finished-game records kept on the device; 0 = 2 * n_games + 64 */
    double c_puct, noise_eps, dirichlet_alpha, tau_decay_rate, resign_threshold, enable_resign_rate;
    uint64_t seed;                   /* counter-based RNG key (Philox4x32-10): u(seed, game_id, stream, index) */
    int32_t use_history;             /* 28 input planes (CChessPlayer(use_history=True), static_env.py:158-194) */
    int32_t reserved;
} cz_search_cfg;

/* finished-game record in the ring: this header, then uint16 moves[max_plies + 2] (labels, mover frame) */
typedef struct cz_game_record {
    uint32_t game_id;
    int32_t turns;                   /* number of moves recorded */
    int32_t value;                   /* +1 red won, -1 black won, 0 draw (self_play.py:190-191) */
    uint32_t flags;                  /* bit 0: store (self_play.py:194-200), bit 1: ended by resignation */
} cz_game_record;

int cz_search_create(const cz_search_cfg* cfg, cz_search** out);   /* allocates device memory on the current device */
int cz_search_destroy(cz_search* s);
size_t cz_search_bytes(const cz_search* s);
/* out[16]: G, K, sims, pool chunks, chunk-table entries per game, hash_cap, max_depth, max_plies, record_stride,
 * ring_cap, n_counters, input planes (14 / 28), chunks a game always keeps, longest no_act list, 0, 0 */
int cz_search_info(const cz_search* s, int32_t* out);
/* HOST out[8]: pool chunks, free chunks, chunks owned by games, ... by the largest game, tree bytes in use, ... of the
 * largest game, nodes in all trees, ... in the largest tree; synchronises the stream.
 * Tree memory: a game's tree is kept for the whole game (the reference's behaviour) in 1 MiB chunks taken from a pool
 * shared by all games; only when a ply cannot be reserved (pool empty) is that game's tree dropped -- counter
 * tree_resets; simulations that still find no room end with value 0 -- counter overflow_sims. */
int cz_search_memory_info(cz_search* s, int64_t* host_out, void* stream);

/* self-play mode: every slot plays games from INIT_STATE forever; slot g starts with game id
 * first_game_id + g and continues with + game_id_stride after each finished game (0 = n_games). */
int cz_search_start_selfplay(cz_search* s, uint64_t seed, uint32_t first_game_id, uint32_t game_id_stride, void* stream);

/* external mode (CChessPlayer.action): set the position to search for each game.  boards [G][90];
 * turns [G] or NULL; no_act [G][32] + n_no_act [G] or NULL (at most 32 banned moves per game); increase_temp / enable_resign [G] or NULL;
 * select_mask [G] or NULL (only games with a non-zero byte are touched).  Trees are kept (subtree reuse).
 * use_history only: hist_kind [G] or NULL = the `hist` argument of action(): 0 none, 1 prev_boards[g] ([G][90]) is
 * the game position two plies before the root, 2 a history shorter than 5 entries was passed. */
int cz_search_set_roots(cz_search* s, const int8_t* boards, const int32_t* turns, const uint16_t* no_act,
                        const uint8_t* n_no_act, const uint8_t* increase_temp, const uint8_t* enable_resign,
                        const uint8_t* select_mask, const int8_t* prev_boards, const uint8_t* hist_kind, void* stream);

/* one lock-step round for all games.  policy [G*K][2086] float32, value [G*K] float32 (results for the
 * planes written by the previous round; ignored for slots that had no leaf), planes [G*K][14 or 28][10][9]. */
int cz_search_round(cz_search* s, const float* policy, const float* value, void* planes, void* stream);

/* The same round with a COMPACT evaluation queue: only the slots that hold a new leaf are evaluated.  After the round
 * q_rows[0 .. *q_count) (int32 DEVICE, arbitrary order) lists those slots and *q_count (int32 DEVICE) their number; the
 * caller evaluates planes[q_rows[i]] and writes the result to policy[i] / value[i] -- row i, not the slot -- which the
 * NEXT cz_search_round_q consumes.  Nothing is copied to the host: run the network with the cz_*_q entry points, which
 * read the board count from q_count on the device (fixed launch shapes: the round still replays from a HIP graph).
 * The two forms may be mixed: a round consumes its results by compact row exactly when the PREVIOUS round of the
 * object was a cz_search_round_q.  In self-play 2-7 % of the slots carry no leaf (terminal / repeated positions, parked simulations), with
 * search_threads = 32-40 up to half of them. */
int cz_search_round_q(cz_search* s, const float* policy, const float* value, void* planes, int32_t* q_rows,
                      int32_t* q_count, void* stream);

/* simulations per search for the following cz_search_set_roots calls (CChessPlayer.action(depth=...), player.py:160) */
int cz_search_set_sims(cz_search* s, int simulation_num_per_move);
/* on = 1: the policy rows handed to cz_search_round(_q) are raw LOGITS, not probabilities.  The reference spreads the
 * softmax output over the legal moves, p_j / sum_legal p (agent/player.py:272-283); the softmax's own denominator cancels
 * there, so the priors are formed as exp(l_j - max over the node's moves) / their sum -- identical up to float32 rounding,
 * and the network's tail can skip normalising all 2086 columns (cz_heads_tail normalize = 0).  Default 0. */
int cz_search_policy_logits(cz_search* s, int on);
/* (round 5) masks [n_games * sims_per_round][96] uint32 DEVICE, caller-owned (NULL switches it off, the default): every new
 * leaf's position is ALSO written as an occupancy board into row `slot` -- word pos = plane position, bit c = plane c (0..13 the
 * position, 14..27 the history block of 28-plane searches) shows a piece there; state_to_planes / state_history_to_planes,
 * environment/static_env.py:137-194, in 384 bytes -- for cz_input_resblock_m.  The planes are written as before. */
int cz_search_leaf_masks(cz_search* s, uint32_t* masks);
/* (round 5) on = 0: while cz_search_leaf_masks is set, a new leaf is written as its occupancy board ONLY -- for a caller whose
 * network takes the boards (cz_input_resblock_m reads nothing else); the `planes` rows of cz_search_round(_q) are then left
 * untouched.  on = 1 (default) writes both.  CZ_ERR_ARG when switched off without masks; clearing the masks switches the
 * planes back on.  (state_to_planes, environment/static_env.py:137-156: the same information in 384 bytes.) */
int cz_search_leaf_planes(cz_search* s, int on);
int cz_search_reset_trees(cz_search* s, void* stream);
/* synchronises the stream; *host_out = number of games whose current search is unfinished */
int cz_search_pending(cz_search* s, int* host_out, void* stream);
/* After a cz_search_round: the queue rows (slot = game * K + sim) that hold a NEW leaf, i.e. the only rows whose
 * policy / value the next round will read.  rows [G*K] int32 DEVICE (compacted, arbitrary order), counts_dev [2] int32
 * DEVICE scratch; HOST host_out[0] = searches still running (as cz_search_pending), host_out[1] = rows written.
 * Synchronises the stream.  Lets a caller with few games (arena, UCI) evaluate only the rows that carry a position:
 * with K = 32 simulations per batch roughly half of a batch's rows are simulations parked on a leaf another one
 * already expanded (player.py:238-242). */
int cz_search_leaf_rows(cz_search* s, int32_t* rows, int32_t* counts_dev, int* host_out, void* stream);
/* root edges after a search: moves/n/w/p [G][128], sum_n [G], counts [G] (any may be NULL) */
int cz_search_root_stats(cz_search* s, uint16_t* moves, int32_t* n, double* w, float* p, int32_t* sum_n,
                         uint8_t* counts, void* stream);
/* the same for the node reached from each root along path[g][0 .. path_len) (DEVICE move labels, 0xFFFF ends a path
 * early; path_len = 0: the root).  A position that is not linked in the tree reports counts = 0.  This is what the
 * reference's callers read out of the search_tree dict they handed to the player (ponder move uci.py:312-318,
 * principal variation player.py:408-450). */
int cz_search_node_stats(cz_search* s, const uint16_t* path, int path_len, uint16_t* moves, int32_t* n, double* w,
                         float* p, int32_t* sum_n, uint8_t* counts, void* stream);
/* principal variation of every game in one launch (player.py:408-433: the most-visited edge at each node, `>=` keeps
 * the last maximum, the bans of the current search apply at the root): moves [G][max_len] uint16 (0xFFFF padded),
 * visits [G][max_len] int32, both DEVICE. */
int cz_search_pv(cz_search* s, int max_len, uint16_t* moves, int32_t* visits, void* stream);
/* stop starting simulations in every running search (UCI `stop`, CChessPlayer.close_and_return_action,
 * player.py:88-106): the next cz_search_round backs up what is in flight and the searches become idle */
int cz_search_stop(cz_search* s, void* stream);
/* calc_policy + apply_temperature + np.random.choice with the uniform draws u [G] (NULL = 0.5):
 * action [G] = label, or -1 when the player resigns */
int cz_search_choose(cz_search* s, const double* u, int32_t* action, void* stream);
/* HOST out[n_counters] (order: enum Counter in csrc/xq_search.h); synchronises the stream */
int cz_search_counters(cz_search* s, uint64_t* host_out, void* stream);
/* the same counters before the sum over games: HOST out[G][n_counters] (tuning: which game's wavefront a launch waits for,
 * tools/search_tail.py); synchronises the stream */
int cz_search_game_counters(cz_search* s, uint64_t* host_out, void* stream);
/* copies finished-game records (record_stride bytes each) written since *cursor into HOST memory */
int cz_search_drain_records(cz_search* s, unsigned int* cursor, void* host_buf, int max_records, int* n_out,
                            void* stream);
/* ---- network epilogue -----------------------------------------------------------------------------
 * x = relu?(x + bias[c] (+ residual)) in place over a channels-last activation x[rows][channels]
 * (n_elems = rows * channels, channels % 8 == 0, dtype CZ_F32 / CZ_F16 / CZ_BF16).  Replaces the separate
 * bias / add / ReLU passes that follow each trunk convolution of agent/model.py:68-83 (BatchNorm folded). */
int cz_bias_act(void* x, const void* bias, const void* residual, size_t n_elems, int channels, int dtype,
                int relu, void* stream);

/* ---- trunk convolution (hand-written MFMA kernel, csrc/xq_conv.hip) -------------------------------------------
 * Replaces Conv2D(F, 3, padding="same") -> BatchNorm -> (+ skip) -> ReLU of the residual tower
 * (agent/model.py:40-83, BatchNorm folded into w / bias) on channels-last 10x9 boards:
 *   y[n][pix][o] = act( sum_{ky,kx,c} w[o][c][ky][kx] * x[n][pix + (ky-1, kx-1)][c] + bias[o] (+ skip[n][pix][o]) )
 * Activations are [n_boards][90][channels] arrays of 2-byte elements (dtype CZ_BF16 or CZ_F16); fp32 accumulate.
 *   parts = 1: plain bf16 / fp16 operands (x_lo, skip_lo, y_lo unused).
 *   parts = 2: "split" operands -- every value is a pair hi + lo (lo = value - hi rounded again) and the kernel
 *              accumulates hi*hi + hi*lo + lo*hi: fp32-class results (product error ~2^-17) from three
 *              bf16 MFMAs.  Outputs are re-split into (y_hi, y_lo).
 * y_f32 != NULL: write the fp32 result there instead of y_hi / y_lo (last trunk layer, feeds the heads).
 * w_packed: device copy of what cz_conv3x3_pack_weights produced for the same channels / dtype / parts.
 * channels in {32, 128, 192, 256}.  skip_hi may be NULL (no residual). */
int cz_conv3x3(const void* x_hi, const void* x_lo, const void* w_packed, const float* bias, const void* skip_hi,
               const void* skip_lo, void* y_hi, void* y_lo, float* y_f32, int n_boards, int channels, int dtype,
               int parts, int relu, void* stream);
/* A whole residual block in one launch (csrc/xq_conv.hip, k_resblock):
 *   y = relu( conv3x3(relu(conv3x3(x, w1) + bias1), w2) + bias2 + x )          (agent/model.py:68-83)
 * The intermediate activation and the skip operand stay in LDS; HBM sees one read of x and one write of y.
 * Same layouts and `parts` as cz_conv3x3; y_f32 != NULL (parts = 2 only) writes the fp32 result instead of
 * (y_hi, y_lo); y may alias x.  Supported: 128 filters (parts 1 or 2), 192 / 256 filters (parts 1); anything else returns
 * CZ_ERR_ARG (use two cz_conv3x3 calls).  Bit-identical to the two-call form.
 * dtype CZ_F16C8 (128 or 192 filters, parts = 2): the c8 arithmetic -- x_lo / y_lo are c8 images, the filters are
 * cz_conv3x3_c8_pack_weights' (see cz_conv3x3_c8 below); bit-identical to two cz_conv3x3_c8 calls (k_resblock_c8,
 * k_resblock_ip_c8).  cz_input_conv (filters packed with CZ_F16, parts 2; u8 or fp32 planes) and the _q forms take the
 * same code at both filter counts, cz_resblock_heads and cz_input_resblock at 128 filters.
 * dtype CZ_F16 with parts = 2 ("f16x3": (hi, lo) fp16 pairs, 22 bits per operand) is the more exact sibling of the bf16
 * pairs at the same cost; it needs activations and folded filters inside fp16's range. */
int cz_resblock(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1, const void* w2_packed,
                const float* bias2, void* y_hi, void* y_lo, float* y_f32, int n_boards, int channels, int dtype,
                int parts, void* stream);
/* The c8 tower arithmetic (csrc/xq_conv.hip, k_conv3x3_c8 / k_resblock_c8 / k_resblock_ip_c8, K loop csrc/xq_c8_kloop.h;
 * DESIGN section 7b; what the self-play engine requested by default until the end of round 4 -- now its bf6 sibling c6, below, with c8
 * next in the guard's chain -- and keeps where its load-time check against float64
 * allows -- a host binding these entry points directly owns that check): one 3x3 convolution, 128 or 192
 * filters (the shapes below are for 128), computed as  f16(w) f16(x) + e4m3(w) e4m3(x - f16(x)) + e4m3(w - f16(w)) e4m3(x)  (one fp16 and two block-scaled
 * fp8 matrix instructions per 64 input channels instead of three bf16 ones).  x_hi: f16 [n][90][128]; x_c8: bytes
 * [n][90][256] = e4m3(x_lo * 2^11) for the 128 channels, then e4m3(x) for them; y = conv + bias (+ skip pair) (ReLU if
 * relu), written as fp32 (y_f32) or as the operand pair (y_hi, y_c8).
 * The accumulators start at bias (+ the skip pair's value) and the products are added on top (round 4).
 * Reference arithmetic: Keras float32 (agent/model.py:32-83); per-product accuracy ~2^-16 (the split-bf16 form: 2^-17; the
 * fp16 pairs: 2^-21), two thirds of their matrix-pipe time.  Activations above 448 lose the w_lo x correction (e4m3
 * saturates): scale the folded network by a power of two first (agent/model.py choose_act_shift). */
/* Packed filter = fp16 fragments, correction fragments, 4 ints, then 2 x channels signed bytes: the correction operands'
 * power-of-two shifts PER OUTPUT CHANNEL (e4m3(w 2^s) with the row's largest magnitude in [128, 256), the same for
 * w - f16(w); c6: bf6, [8, 16)) -- the matrix instruction takes the filter operand's scale per row, so channels of very
 * different magnitude (folded BatchNorm scales) all keep their correction precision. */
size_t cz_conv3x3_c8_packed_bytes(int channels);
int cz_conv3x3_c8_pack_weights(const float* w_oihw, int channels, void* out_host);
int cz_conv3x3_c8(const void* x_hi, const void* x_c8, const void* w_packed, const float* bias,
                  const void* skip_hi, const void* skip_c8, void* y_hi, void* y_c8, float* y_f32,
                  int n_boards, int channels, int relu, void* stream);

/* The c6 tower arithmetic (round 4; k_resblock_c8<.., C6>, 128 filters, whole residual blocks only): the c8 sum with the
 * two correction operands in bf6 (e3m2) -- v_mfma_scale_f32_32x32x64_f8f6f4 retires bf6 operands in 32 cycles, e4m3 ones in
 * 64 (tools/probes/bf6_probe.hip), so a product costs 1.5 instead of 2.0 MFMA-equivalents; per-product accuracy ~2^-15.
 * Activation images keep the c8 pair's shape (x_hi f16 [n][90][128], image bytes [n][90][256]); the image holds, per pixel
 * and 32-channel block, a 24-byte piece bf6((x - f16(x)) 2^(11 - k)) and a piece bf6(x 2^-k) (layout: csrc/xq_conv.hip,
 * namespace rb8), k = the image's exponent, 2^k * 28 >= max |x| (from calibration activations; the conversion saturates).
 * Filters: cz_conv3x3_c6_pack_weights(w, 128, x_exp, y_exp, out) -- same size as cz_conv3x3_c8_packed_bytes(128) -- with
 * the exponents of the image the convolution READS and of the one it WRITES; a block's w1 / w2 must agree on the
 * intermediate image's exponent, consecutive blocks on the stream image's.  Entry points (dtype CZ_F16C6): cz_resblock,
 * cz_resblock_heads, cz_input_resblock (whose gathered input image is c8: ITS w1 is cz_conv3x3_c8_pack_weights') and the
 * _q forms.  A host owns the accuracy check exactly as for c8 (agent/model.py guarded_inference_net: c6 -> c8 -> ...).
 * y_exp = 127 on a block's SECOND filter: the block writes a c8 image (y_lo of cz_resblock / cz_input_resblock is then read by
 * c8 blocks) -- the hand-over of a hybrid "c6>N" tower, whose first N blocks run c6 and the rest c8. */
int cz_conv3x3_c6_pack_weights(const float* w_oihw, int channels, int x_exp, int y_exp, void* out_host);

/* test / tuning hook: the 128-filter split residual block with operand-pair output has two schedules that give
 * bit-identical results -- k_resblock_pipe (default, 1): epilogue 2 of a board runs under the next board's first K
 * loop; k_resblock (0).  enable < 0 only queries.  Returns the previous setting. */
int cz_resblock_pipelined(int enable);
/* The LAST residual block of the tower with the two 1x1 head convolutions folded into its store pass (cz_resblock +
 * cz_head_convs in one launch; the block's activation never reaches HBM): split operands, 128 filters,
 * n_policy + n_value == 6.  Outputs as cz_head_convs. */
int cz_resblock_heads(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1,
                      const void* w2_packed, const float* bias2, const float* head_w, const float* head_b,
                      float* policy_feat, float* value_feat, int n_boards, int channels, int dtype, int n_policy,
                      int n_value, void* stream);
/* The input layer AND the first residual block in one launch (128 filters; split operands, or dtype CZ_F16C8 with the c8
 * pair as output and cz_conv3x3_c8_pack_weights filters: k_resblock_c8<FIRST>): the 5x5 input convolution
 * (Conv2D(F, 5, "same") -> BatchNorm -> ReLU, agent/model.py:36-39) of the one-hot feature planes is a gather over the
 * occupied squares, computed in exact fp32 by the block's copy waves while its matrix waves run the previous board.
 *   planes_u8   [n_boards][in_planes][90] uint8, 0 / 1 (what the search kernel writes; in_planes 14 or 28)
 *   in_table    DEVICE fp32 [in_planes][25][128]: in_table[c][ky * 5 + kx][o] = w[o][c][ky][kx] (BatchNorm folded)
 *   in_bias     DEVICE fp32 [128]
 *   rows/n_dev  compact evaluation queue (may be NULL): board i = planes_u8[rows[i]], min(n_boards, *n_dev) boards
 * The rest as cz_resblock with operand-pair output.  Equal to cz_input_conv followed by cz_resblock up to the rounding of
 * the input layer (fp32 sums here, split-bf16 products there). */
int cz_input_resblock(const void* planes_u8, int in_planes, const float* in_table, const float* in_bias,
                      const void* w1_packed, const float* bias1, const void* w2_packed, const float* bias2, void* y_hi,
                      void* y_lo, int n_boards, int channels, int dtype, const int32_t* rows, const int32_t* n_dev,
                      void* stream);
/* (round 5) The same with the positions' OCCUPANCY BOARDS handed in: masks [n_boards or slots][96] uint32 DEVICE, word pos =
 * plane position i * 9 + j (words 90 .. 95 zero), bit c = plane c shows a piece there -- the planes' content in 384 bytes, which
 * is what the block's copy waves otherwise derive from the 1260 (2520) plane bytes before they can start the gather.
 * cz_search_leaf_masks() makes the search kernel write them beside the planes of every new leaf.  masks = NULL: exactly
 * cz_input_resblock; with masks, planes_u8 is not read (and may be NULL).  rows index both arrays alike. */
int cz_input_resblock_m(const void* planes_u8, const uint32_t* masks, int in_planes, const float* in_table,
                        const float* in_bias, const void* w1_packed, const float* bias1, const void* w2_packed,
                        const float* bias2, void* y_hi, void* y_lo, int n_boards, int channels, int dtype,
                        const int32_t* rows, const int32_t* n_dev, void* stream);
/* number of 2-byte elements of the packed filter (all parts, including the prefetch padding); 0 = bad argument */
size_t cz_conv3x3_packed_elems(int channels, int parts);
/* HOST: w_oihw[channels][channels][3][3] fp32 -> MFMA fragment order, split into parts; out_host holds
 * cz_conv3x3_packed_elems() elements */
int cz_conv3x3_pack_weights(const float* w_oihw, int channels, int dtype, int parts, void* out_host);
/* The input convolution of the network (csrc/xq_conv.hip, k_input_conv): Conv2D(F, 5, padding="same") -> BatchNorm ->
 * ReLU on the feature planes (agent/model.py:36-39), BatchNorm folded.  planes: [n_boards][in_planes][10][9] exactly as
 * cz_encode / the search kernel write them (planes_dtype CZ_F32 / CZ_F16 / CZ_BF16 / CZ_U8, in_planes 14 or 28);
 * output: the [n_boards][90][channels] operand (pair) the residual tower reads (dtype CZ_BF16 / CZ_F16, parts as in
 * cz_conv3x3 -- the 0/1 planes are exact in 2 bytes, so parts = 2 splits only the weights). */
int cz_input_conv(const void* planes, int planes_dtype, int in_planes, const void* w_packed, const float* bias,
                  void* y_hi, void* y_lo, int n_boards, int channels, int dtype, int parts, int relu, void* stream);
size_t cz_input_conv_packed_elems(int channels, int in_planes, int parts);
/* Compact-queue forms of the three kernels above (cz_search_round_q): the number of boards is min(n_boards, *n_dev)
 * with n_dev in DEVICE memory (n_boards = the capacity of the buffers = the launch shape), and the input convolution
 * reads board i from planes[rows[i]] (rows DEVICE int32, NULL = identity).  rows / n_dev may be NULL: then exactly the
 * plain function.  Threading: rows / n_dev travel to the launch through thread-local state inside the call, so each
 * call is self-contained on its thread; like every cz_* entry point these may be called from several host threads at
 * once as long as each thread uses its own stream. */
int cz_input_conv_q(const void* planes, int planes_dtype, int in_planes, const void* w_packed, const float* bias,
                    void* y_hi, void* y_lo, int n_boards, int channels, int dtype, int parts, int relu,
                    const int32_t* rows, const int32_t* n_dev, void* stream);
int cz_resblock_q(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1, const void* w2_packed,
                  const float* bias2, void* y_hi, void* y_lo, float* y_f32, int n_boards, int channels, int dtype,
                  int parts, const int32_t* n_dev, void* stream);
/* (round 5) n_blocks (2 .. 8) CONSECUTIVE c6 residual blocks of a 128-filter tower in ONE launch (k_tower_c6): the same
 * arithmetic as n_blocks calls of cz_resblock(dtype CZ_F16C6) -- bit-identical results -- with the activations staying in the
 * CU's LDS between the blocks (a workgroup takes a pair of boards through the whole chain; HBM sees a board at the chain's
 * entry and exit only).  Replaces the inner part of the residual tower, agent/model.py:41-43 (`for _ in range(res_layer_num):
 * x = self._build_residual_block(x)`).  w1_packed / bias1 / w2_packed / bias2: HOST arrays of n_blocks DEVICE pointers
 * (cz_conv3x3_c6_pack_weights filters; every block's second filter carries a c6 output exponent, block b + 1 reads the image
 * block b writes).  x / y: c6 operand pairs [n_boards][90][128] f16 + [n_boards][90][256] bytes.  n_dev: compact queue (DEVICE
 * int32, may be NULL): min(n_boards, *n_dev) boards. */
int cz_tower_c6(const void* x_hi, const void* x_c6, int n_blocks, const void* const* w1_packed, const float* const* bias1,
                const void* const* w2_packed, const float* const* bias2, void* y_hi, void* y_c6, int n_boards,
                const int32_t* n_dev, void* stream);
/* The same chain ending on the tower's LAST block with the 1 x 1 head convolutions as its exit (cz_resblock_heads' outputs:
 * policy_feat [n][n_policy * 90], value_feat [n][n_value * 90] fp32, ReLU'd; agent/model.py:46-60, the heads' first layers).  The
 * head dot products are summed over a pixel's four 32-channel partial sums: equal to cz_resblock_heads up to the rounding of
 * that summation order (2e-6 relative). */
int cz_tower_c6_heads(const void* x_hi, const void* x_c6, int n_blocks, const void* const* w1_packed, const float* const* bias1,
                      const void* const* w2_packed, const float* const* bias2, const float* head_w, const float* head_b,
                      float* policy_feat, float* value_feat, int n_boards, int n_policy, int n_value, const int32_t* n_dev,
                      void* stream);
/* (round 6) The chain for EVERY tower arithmetic.  A chain block's two images (the one its first convolution reads = the one the
 * block before wrote, and its intermediate image) each have a format: */
#define CZ_IMG_C8 0      /* f16 + e4m3 corrections (cz_conv3x3_c8_pack_weights filters read it) */
#define CZ_IMG_C6 1      /* f16 + bf6 pieces with an exponent (cz_conv3x3_c6_pack_weights) */
#define CZ_IMG_PAIR 2    /* (hi, lo) fp16 / bf16 pair (cz_conv3x3_pack_weights, parts = 2) */
#define CZ_EXIT_HEADS 3  /* exit only: the 1 x 1 head convolutions instead of an image */
/* cz_tower: n_blocks (1 .. 8) consecutive residual blocks on the c8 OR the c6 arithmetic in ONE launch -- bit-identical to
 * n_blocks calls of cz_resblock with the matching dtype.  Kernel (end of round 6): k_resblock_ip4_c8<128> -- a pair of boards
 * per workgroup with one LDS image each, both epilogues in place, FOUR matrix waves of two channel tiles (a pixel fragment from
 * LDS feeds two MFMAs, no copy waves); environment CZ_TOWER4=0: k_tower (the pair alternating through X | X | Y with copy waves
 * converting the staged result) -- same bits, 6 % slower in the engine.  fmt_x[b] / fmt_y[b] (HOST int arrays; NULL = all CZ_IMG_C6): the
 * format of the image block b's first / second filter reads -- one format per chain (all CZ_IMG_C8 or all CZ_IMG_C6; a hybrid
 * tower is one chain per arithmetic).  exit_fmt: what the last block's result becomes -- CZ_IMG_C6 / CZ_IMG_C8: that operand
 * pair in (y_hi, y_img) (a c6 chain whose last block carries y_exp = 127 ends on CZ_IMG_C8: the hand-over of a "c6>N" tower);
 * CZ_IMG_PAIR (c8 chains): (hi, lo) fp16 pairs, y_img = the lo array [n][90][128] f16: the hand-over of a "c8>N" tower to its
 * f16x3 blocks (cz_resblock's y_f32 + cz_split_bias_act in one); CZ_EXIT_HEADS: the head features (cz_resblock_heads'
 * outputs; y_hi / y_img unused).  Replaces agent/model.py:41-43 for those blocks. */
int cz_tower(const void* x_hi, const void* x_img, int n_blocks, const void* const* w1_packed, const float* const* bias1,
             const void* const* w2_packed, const float* const* bias2, const int* fmt_x, const int* fmt_y, int exit_fmt,
             void* y_hi, void* y_img, const float* head_w, const float* head_b, float* policy_feat, float* value_feat,
             int n_policy, int n_value, int n_boards, const int32_t* n_dev, void* stream);
/* cz_tower_pairs: the same for (hi, lo) pair blocks (f16x3 / bf16x3 arithmetic, dtype CZ_F16 / CZ_BF16; k_tower_pairs):
 * bit-identical to n_blocks calls of cz_resblock(parts = 2).  head_w != NULL: the chain ends on the tower's last block and
 * writes the head features (from hi + lo of the block's result) instead of (y_hi, y_lo). */
int cz_tower_pairs(const void* x_hi, const void* x_lo, int n_blocks, const void* const* w1_packed, const float* const* bias1,
                   const void* const* w2_packed, const float* const* bias2, void* y_hi, void* y_lo, const float* head_w,
                   const float* head_b, float* policy_feat, float* value_feat, int n_policy, int n_value, int n_boards,
                   int dtype, const int32_t* n_dev, void* stream);
/* cz_resblock_chain (round 6): n_blocks (1 .. 12) consecutive residual blocks of a 192-FILTER tower (the reference's deployed
 * width, configs/distribute.py:84-87) on one staged arithmetic -- dtype CZ_F16C8, or CZ_F16C6 (c6 blocks behind the tower's first
 * one, which reads the input layer's c8 image: cz_resblock with CZ_F16C86) -- in ONE launch: a workgroup takes a PAIR of boards
 * through all blocks, one LDS image per board (both epilogues in place), on four matrix waves of three channel tiles each -- six
 * channel tiles spread evenly over the CU's four SIMDs (k_resblock_ip4_c8; environment CZ_IP_PAIR=0: one board in two images on
 * six matrix waves, k_resblock_ip_c8).  HBM sees a board at the entry and the exit.  Bit-identical to n_blocks calls of cz_resblock.  y_f32 != NULL: the last block writes fp32 [n][90][192]
 * instead of the operand pair (the tower's last block / the hand-over of a c8>N tower).  dtype CZ_F16C86: a c6 chain that STARTS
 * the tower -- its block 0 reads the input layer's c8 image (first filter c8-packed, as for cz_resblock with CZ_F16C86), the
 * blocks behind it are c6 blocks: a 10 x 192 c6 tower is one launch.  dtype CZ_F16 / CZ_BF16: (hi, lo) PAIR
 * blocks (f16x3 / bf16x3 -- what the load-time guard gives a peaked-policy network at this width), x_img / y_img = the lo tensors
 * [n][90][192]; same shape of kernel (k_tower_pairs4<E, 192>). */
int cz_resblock_chain(const void* x_hi, const void* x_img, int n_blocks, const void* const* w1_packed, const float* const* bias1,
                      const void* const* w2_packed, const float* const* bias2, void* y_hi, void* y_img, float* y_f32, int n_boards,
                      int channels, int dtype, const int32_t* n_dev, void* stream);
/* cz_tower_plain (round 6): n_blocks (1 .. 24) consecutive residual blocks of a 256-FILTER tower on plain fp16 / bf16 operands
 * (dtype CZ_F16 / CZ_BF16, cz_conv3x3_pack_weights with parts = 1) in ONE launch -- BASELINE configs[4], the 20 x 256 fp16 tower,
 * is a single launch: a workgroup takes a PAIR of boards through all blocks with ONE LDS image per board (a filter fragment
 * feeds both boards; the intermediate activation overwrites the block's input, whose values wait as the skip operand in
 * registers / spare LDS; environment CZ_TOWER_PLAIN_PAIR=0: one board in two images).  Bit-identical to n_blocks calls of
 * cz_resblock(parts = 1).  x / y: [n_boards][90][256]. */
int cz_tower_plain(const void* x, int n_blocks, const void* const* w1_packed, const float* const* bias1,
                   const void* const* w2_packed, const float* const* bias2, void* y, int n_boards, int channels, int dtype,
                   const int32_t* n_dev, void* stream);
int cz_resblock_heads_q(const void* x_hi, const void* x_lo, const void* w1_packed, const float* bias1,
                        const void* w2_packed, const float* bias2, const float* head_w, const float* head_b,
                        float* policy_feat, float* value_feat, int n_boards, int channels, int dtype, int n_policy,
                        int n_value, const int32_t* n_dev, void* stream);
/* HOST: w_oihw[channels][in_planes][5][5] fp32 -> MFMA fragment order (cz_input_conv_packed_elems() elements) */
int cz_input_conv_pack_weights(const float* w_oihw, int channels, int in_planes, int dtype, int parts, void* out_host);
/* fp32 activation x[rows][channels] (+ bias[c], may be NULL) -> ReLU? -> (y_hi, y_lo) operand pair (parts = 2) or
 * a plain bf16 / fp16 copy (parts = 1).  Used after the 5x5 input convolution. */
int cz_split_bias_act(const float* x, const float* bias, void* y_hi, void* y_lo, size_t n_elems, int channels,
                      int dtype, int parts, int relu, void* stream);

/* The two 1x1 head convolutions (policy Conv2D(4,1), value Conv2D(2,1), BatchNorm folded, ReLU; agent/model.py:56-63)
 * in one streaming pass over the trunk output x[n_boards][90][channels] (dtype CZ_F32 / CZ_F16 / CZ_BF16):
 *   w[n_policy + n_value][channels] fp32 (policy filters first), bias[n_policy + n_value] fp32;
 *   policy_feat[n_boards][n_policy * 90], value_feat[n_boards][n_value * 90] fp32 in channels-first Flatten order.
 * n_policy + n_value must be 6 (the reference's 4 + 2). */
int cz_head_convs(const void* x, int dtype, const float* w, const float* bias, float* policy_feat, float* value_feat,
                  int n_boards, int channels, int n_policy, int n_value, void* stream);

/* The dense tail of both heads (agent/model.py:58-59 and :64-66: Flatten -> Dense(2086, softmax); Flatten -> Dense(256,
 * relu) -> Dense(1, tanh)) on the head features cz_head_convs / cz_resblock_heads produce, in three launches of
 * hand-written kernels (csrc/xq_heads.hip: split-bf16 MFMA GEMM tiles of 64 positions with the softmax statistics
 * kept per lane, one normalising pass, the value head with its hidden layer in the accumulators).
 *   policy_feat[n][n_policy_feat], value_feat[n][n_value_feat]   fp32 (180 or 360 features each: 2 or 4 head filters)
 *   wp_packed / w1_packed    cz_fc_pack_weights() of the [n_labels][n_policy_feat] / [n_hidden][n_value_feat] matrices
 *   bias_p[n_labels], bias1[n_hidden], w2[n_hidden], b2          fp32 (n_labels even)
 *   policy[n][n_labels] (softmax), value[n] (tanh)               fp32 outputs
 *   stats_scratch            DEVICE scratch of 2 * n_boards floats (the rows' max / sum of exp between the launches)
 *   n_dev                    NULL, or the DEVICE int32 count of the compact evaluation queue: only the first
 *                            min(*n_dev, n_boards) rows are computed and written
 *   dtype                    the element type of the packed pairs (the one given to cz_fc_pack_weights): CZ_BF16 or CZ_F16
 *   normalize                1: policy = softmax (the reference's output).  0: policy keeps the raw LOGITS and the pass over
 *                            all n_labels columns is skipped -- for a queue consumed by a search with
 *                            cz_search_policy_logits(h, 1), which needs the legal moves' entries only
 * Precision: operands as (hi, lo) pairs, three MFMAs per product, fp32 accumulation -- the tower's arithmetic: 2^-17 per
 * product with bf16 pairs, 2^-21 class with fp16 pairs (22 bits per operand; the head features are O(1), well inside
 * fp16's range, and the matrix unit honours fp16 subnormals -- tools/f16x3_probe.py). */
int cz_heads_tail(const float* policy_feat, int n_policy_feat, const void* wp_packed, const float* bias_p,
                  int n_labels, const float* value_feat, int n_value_feat, const void* w1_packed, const float* bias1,
                  int n_hidden, const float* w2, float b2, float* policy, float* value, float* stats_scratch,
                  int n_boards, int dtype, int normalize, const int32_t* n_dev, void* stream);
/* number of 2-byte elements of a packed dense layer (0 = bad argument); HOST: w[n_out][n_in] fp32 -> (hi, lo) pairs of
 * `dtype` (CZ_BF16 / CZ_F16) in fragment order */
size_t cz_fc_packed_elems(int n_out, int n_in);
int cz_fc_pack_weights(const float* w, int n_out, int n_in, int dtype, void* out_host);

/* test hook: out[n] (DEVICE, float64) = n draws of the root noise np.random.dirichlet(alpha * ones(n_moves))[0]
 * (agent/player.py:304) from the generator the search kernel uses (k_noise; csrc/xq_noise.h: counter-based integer
 * hash keyed by seed / game_key, float32 Marsaglia-Tsang Gamma draws) */
int cz_debug_noise(uint64_t seed, uint32_t game_key, double alpha, int n_moves, double* out, int n, void* stream);
/* test hook: y[i] = sqrt((double)(x[i] + 1)) exactly as the PUCT kernel computes it */
int cz_debug_sqrt(const int32_t* x, double* y, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
