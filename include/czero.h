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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CZ_VERSION 1

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

#ifdef __cplusplus
}
#endif
#endif
