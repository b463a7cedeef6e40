// xq_search.h -- device-side data layout of the batched PUCT-MCTS engine (gfx950).
//
// One wavefront per game tree.  Everything a tree owns lives in one per-game slice of a few
// structure-of-arrays buffers in HBM (so that a wave's loads of N/W/P/move for one node are
// contiguous), the hot scratch (boards, ordered move lists, the current path) lives in LDS.
//
// Reference objects replaced (cchess_alphazero/agent/player.py):
//   VisitState (:17-25)  -> node_* arrays          ActionState (:28-33) -> e_* arrays
//   tree = defaultdict keyed by state string (:49) -> per-game open-addressing hash on the packed board
//   history lists of MCTS_search (:198-260)        -> s_path_* per simulation slot
#pragma once
#include <stdint.h>

namespace xq {

constexpr int KEY_WORDS = 12;          // 90 squares x 4 bit, padded to 48 B
constexpr int MAX_NO_ACT = 16;         // banned root moves per ply (self_play.py:161-175)
constexpr int CHILD_UNKNOWN = -1;
constexpr int CHILD_TERM_WIN = -2;     // done() == (True, +1): value for the child's mover +1 -> x2
constexpr int CHILD_TERM_LOSS = -3;    // done() == (True, -1)

enum SimState : uint8_t { SIM_IDLE = 0, SIM_LEAF = 1, SIM_PARKED = 2 };

enum NodeFlag : uint32_t { NODE_WAITING = 1u << 8 };    // low 8 bits of node_meta = move count

enum GameMode : int { MODE_EXTERNAL = 0, MODE_SELFPLAY = 1 };

enum GamePhase : uint8_t {
    PH_IDLE = 0,        // nothing to do (external mode: waiting for set_roots / choose)
    PH_SEARCH = 1,      // simulations outstanding for the current root
    PH_READY = 2,       // search of the current root complete (external mode: results can be read)
    PH_COMPACT = 3,     // self-play: the next search needs the arena compacted first (k_compact, off the round's critical path)
    PH_COMPACTED = 4,   // k_compact is done; the NEXT round's first kernel turns this into PH_SEARCH (see k_compact)
};

// per-game counters (uint64 each); summed on the host
enum Counter : int {
    CT_SIMS = 0, CT_EXPANSIONS, CT_TERMINAL_SIMS, CT_REPETITION_SIMS, CT_PARKED, CT_SUM_DEPTH, CT_MAX_DEPTH,
    CT_EDGES_VISITED, CT_LEAF_MOVES, CT_PLIES, CT_GAMES, CT_RED_WINS, CT_BLACK_WINS, CT_DRAWS, CT_RESIGNS,
    CT_TREE_RESETS, CT_OVERFLOW_SIMS, CT_DEPTH_OVERFLOW, CT_ROOT_REUSED_SIMS, CT_RING_DROPPED, CT_TREE_COMPACTIONS,
    CT_COUNT
};

struct SearchParams {
    // sizes
    int G, K, sims, vl;
    int node_cap, edge_cap, hash_cap, max_depth, max_plies;
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
};

struct SearchBuffers {
    // ---- tree, per game ----
    uint32_t* node_key;     // [G][node_cap][12]
    int32_t* node_sum_n;    // [G][node_cap]
    uint32_t* node_eoff;    // [G][node_cap]  first edge (index inside the game's edge slice)
    uint32_t* node_meta;    // [G][node_cap]  move count | flags
    uint64_t* hash_tab;     // [G][hash_cap]  tag << 32 | node + 1, 0 = empty
    int32_t* e_n;           // [G][edge_cap]
    double* e_w;            // [G][edge_cap]
    float* e_p;             // [G][edge_cap]
    uint16_t* e_mv;         // [G][edge_cap]
    int32_t* e_child;       // [G][edge_cap]
    int32_t* g_node_count;  // [G]
    int32_t* g_edge_count;  // [G]
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
};

// finished-game record header (followed by uint16 moves[max_plies + 2])
struct GameRecord {
    uint32_t game_id;
    int32_t turns;
    int32_t value;      // from red's view: +1 red won, -1 black won, 0 draw (self_play.py:190-191)
    uint32_t flags;     // bit 0 store, bit 1 resigned
};

}  // namespace xq
