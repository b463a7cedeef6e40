// xq_search.h -- device-side data layout of the batched PUCT-MCTS engine (gfx950).
//
// One wavefront per game tree.  The reference keeps a game's whole tree for the whole game
// (worker/self_play.py:84,98-100); 4096 games x 800 simulations x ~150 plies is ~130 GB of tree, which is what
// the 288 GB of HBM are for.  Trees live in a POOL of 1 MiB chunks shared by all games of the search object:
//   * a game owns a list of chunks (g_chunk_tab) and bump-allocates variable-size records in them; a record is
//     addressed by a 32-bit id = local chunk number << 16 | 16-byte granule inside the chunk;
//   * NODE record  = 48 B packed position | 16 B header {sum_n, move count | flags, stat id, -} | float p[nm] |
//     uint16 move[nm]: everything a wave needs to price a node's edges is one contiguous run;
//   * STAT block   = nm x 16 B {double W, int32 N, int32 child}: allocated only when a node is first selected
//     FROM (two nodes in three are leaves that never are), edge j of a node = granule stat + j, so an edge id is a
//     record id as well and a path entry is one word;
//   * chunks are taken from / returned to a device-side free ring (pool_* below) between plies and between games,
//     never inside the simulation kernel: begin_search reserves the worst case of the coming ply.
// The hot scratch (boards, ordered move lists, the current path, the game's chunk table) lives in LDS.
//
// Reference objects replaced (cchess_alphazero/agent/player.py):
//   VisitState (:17-25)  -> node record          ActionState (:28-33) -> p / move in the node record + stat block
//   tree = defaultdict keyed by state string (:49) -> per-game open-addressing hash on the packed board
//   history lists of MCTS_search (:198-260)        -> s_path_* per simulation slot
#pragma once
#include <stdint.h>

namespace xq {

constexpr int KEY_WORDS = 12;          // 90 squares x 4 bit, padded to 48 B
constexpr int MAX_NO_ACT = 32;         // banned root moves per ply (self_play.py:161-175); longer lists are refused by the host
constexpr int CHUNK_SHIFT = 16;        // granules (16 B) per chunk = 65536 -> 1 MiB chunks
constexpr int CHUNK_GRANULES = 1 << CHUNK_SHIFT;
constexpr size_t CHUNK_BYTES = (size_t)CHUNK_GRANULES * 16;
constexpr int MAX_CHUNKS = 256;        // per game (LDS copy of the chunk table): 256 MiB of tree per game at most
constexpr int NODE_HDR_GRANULES = 4;   // key (3) + header (1)
constexpr int RESERVE_MOVES = 80;      // moves per new node the per-ply reservation assumes (observed maximum 70)
constexpr int CHILD_UNKNOWN = -1;
constexpr int CHILD_TERM_WIN = -2;     // done() == (True, +1): value for the child's mover +1 -> x2
constexpr int CHILD_TERM_LOSS = -3;    // done() == (True, -1)

enum SimState : uint8_t { SIM_IDLE = 0, SIM_LEAF = 1, SIM_PARKED = 2,
                          SIM_DEFERRED = 3 };   // ended on a terminal / repeated position; backed up when the phase's descents are over (never outlives a launch)

enum NodeFlag : uint32_t { NODE_WAITING = 1u << 8 };    // low 8 bits of node_meta = move count

enum GameMode : int { MODE_EXTERNAL = 0, MODE_SELFPLAY = 1 };

enum GamePhase : uint8_t {
    PH_IDLE = 0,        // nothing to do (external mode: waiting for set_roots / choose)
    PH_SEARCH = 1,      // simulations outstanding for the current root
    PH_READY = 2,       // search of the current root complete (external mode: results can be read)
};

// per-game counters (uint64 each); summed on the host
enum Counter : int {
    CT_SIMS = 0, CT_EXPANSIONS, CT_TERMINAL_SIMS, CT_REPETITION_SIMS, CT_PARKED, CT_SUM_DEPTH, CT_MAX_DEPTH,
    CT_EDGES_VISITED, CT_LEAF_MOVES, CT_PLIES, CT_GAMES, CT_RED_WINS, CT_BLACK_WINS, CT_DRAWS, CT_RESIGNS,
    CT_TREE_RESETS, CT_OVERFLOW_SIMS, CT_DEPTH_OVERFLOW, CT_ROOT_REUSED_SIMS, CT_RING_DROPPED, CT_CHUNKS_TAKEN,
    CT_STAT_BLOCKS,
    CT_NO_ACT_TRUNCATED,    // self-play: a perpetual-check ban that did not fit the MAX_NO_ACT list of the ply (dropped, counted)
#ifdef CZ_SIM_PROFILE       // tuning build (tools/ab_search.sh): shader-clock cycles of wave time per section of a simulation
    CT_CYC_SELECT, CT_CYC_RULES, CT_CYC_HASH, CT_CYC_EXPAND, CT_CYC_REP, CT_CYC_ATTACH, CT_CYC_RESUME_LOAD,
    CT_CYC_KERNEL_SELECT, CT_CYC_KERNEL_BACKUP,
#endif
    CT_COUNT
};

struct SearchParams {
    // sizes
    int G, K, sims, vl;
    int hash_cap, max_depth, max_plies;
    int n_chunks;           // chunks in the pool
    int max_chunks;         // chunk-table entries per game (<= MAX_CHUNKS)
    int keep_chunks;        // chunks a game never gives back: enough for one full search on an empty tree
    int max_batches;        // lock-step batches one k_sim launch may start for a game
    int planes_dtype;
    int in_planes;          // 14, or 28 with use_history
    int mode;
    // PUCT / game parameters (reference config.play.*)
    double c_puct;
    float c_puct_f32;
    double noise_eps;
    float one_minus_eps_f32;
    double dirichlet_alpha;
    double tau_decay_rate;
    double resign_threshold;
    int min_resign_turn;
    int evaluate;
    int max_game_length;
    double enable_resign_rate;
    uint64_t seed;
    uint32_t game_id_stride;
    int ring_cap;
    int record_stride;      // bytes per finished-game record
    int policy_logits;      // the network rows hold raw LOGITS (cz_search_policy_logits): priors = exp(l - max over the
                            // node's moves), then the usual renormalisation over them
};

struct SearchBuffers {
    // ---- optional caller-owned output (cz_search_leaf_masks): one 96-word occupancy board per queue slot ----
    uint32_t* leaf_masks;   // [G*K][96] or NULL
    int leaf_planes_off;    // cz_search_leaf_planes(0): with leaf_masks set, the planes of a new leaf are NOT written
    // ---- trees ----
    char* pool;             // [n_chunks] x 1 MiB
    uint32_t* pool_ring;    // [n_chunks] free chunk numbers; entries [head, tail) are free
    unsigned int* pool_head;     // [1] chunks ever taken      (monotonic; ring index = value % n_chunks)
    unsigned int* pool_tail;     // [1] chunks ever returned + the initial fill
    unsigned int* pool_tail_vis; // [1] pool_tail as of the last kernel boundary: takers stay below it
    uint32_t* g_chunk_tab;  // [G][max_chunks] pool chunk number of the game's local chunk i
    int32_t* g_nchunks;     // [G]
    uint32_t* g_heap_top;   // [G] next free granule (record id); 1 on an empty tree (id 0 = none)
    uint64_t* hash_tab;     // [G][hash_cap]  tag << 32 | node id + 1, 0 = empty
    int32_t* g_node_count;  // [G]
    // ---- search state, per game ----
    int32_t* g_root;        // [G] node index of the root, -1 = not in the tree yet
    int8_t* g_board;        // [G][96] current root position (int8 board)
    int32_t* g_tasks_left;  // [G] simulations of this ply not yet launched
    int32_t* g_active;      // [G] simulations of the current batch that have not backed up
    uint8_t* g_phase;       // [G]
    int32_t* g_turns;       // [G]
    uint16_t* g_no_act;     // [G][MAX_NO_ACT]
    uint8_t* g_n_no_act;    // [G]
    uint8_t* g_increase_temp;  // [G]
    // ---- simulation slots, per game x K ----
    uint8_t* s_state;       // [G][K]
    int32_t* s_depth;       // [G][K]
    int32_t* s_node;        // [G][K]  leaf / parked node
    int32_t* s_path_node;   // [G][K][max_depth]
    int32_t* s_path_edge;   // [G][K][max_depth]
    // ---- game loop (self-play mode), per game ----
    uint32_t* g_game_id;    // [G]
    uint8_t* g_enable_resign;  // [G]
    int32_t* g_no_eat;      // [G]
    uint32_t* g_hist_key;   // [G][max_plies + 2][12]
    uint64_t* g_hist_hash;  // [G][max_plies + 2] hash of the key (repetition scan: 64 plies per ballot)
    uint16_t* g_hist_act;   // [G][max_plies + 2]
    // ---- outputs ----
    unsigned long long* counters;   // [G][CT_COUNT]
    uint8_t* ring;          // [ring_cap][record_stride]
    unsigned int* ring_tail;       // [1] total records ever written
    int32_t* g_last_action; // [G] external mode: result of choose
    int32_t* pending;       // [1] scratch for cz_search_pending
    double* noise;          // [G][K][128] Dirichlet(alpha)[0] draws for the root edges, refreshed by k_noise
    uint32_t* g_noise_epoch;   // [G]
    int8_t* g_prev_board;   // [G][96] game position two plies before the root (action(hist=...), 28-plane input)
    uint8_t* g_hist_kind;   // [G] 0 no game history, 1 g_prev_board valid, 2 history too short
    int32_t* s_qrow;        // [G][K] compact evaluation-queue row of the slot's leaf (cz_search_round_q), -1 none
};

// finished-game record header (followed by uint16 moves[max_plies + 2])
struct GameRecord {
    uint32_t game_id;
    int32_t turns;
    int32_t value;      // from red's view: +1 red won, -1 black won, 0 draw (self_play.py:190-191)
    uint32_t flags;     // bit 0 store, bit 1 resigned
};

}  // namespace xq
