/*
 * oracle/xq_rules.c -- TEST INFRASTRUCTURE ONLY (see xq_rules.h).
 *
 * Plain-C restatement of cchess_alphazero/environment/static_env.py on the
 * int8[90] board.  Each function cites the reference lines it follows.  The
 * emission ORDER of moves is part of the contract (it decides PUCT tie-breaks
 * and `final_move`), so loops below keep the reference's iteration order.
 */
#include "xq_rules.h"
#include <string.h>
#include <stdlib.h>

static uint8_t  g_from[XQO_NLABELS];
static uint8_t  g_to[XQO_NLABELS];
static uint16_t g_label_of[XQO_NSQ][XQO_NSQ];
static int      g_init_done = 0;

#define SQ(x, y) ((y) * 9 + (x))

/* ---- label table: lookup_tables.py:62-132 ------------------------------- */
static int add_label(int n, int x0, int y0, int x1, int y1)
{
    g_from[n] = (uint8_t)SQ(x0, y0);
    g_to[n] = (uint8_t)SQ(x1, y1);
    g_label_of[SQ(x0, y0)][SQ(x1, y1)] = (uint16_t)n;
    return n + 1;
}

void xqo_init(void)
{
    static const int kn[8][2] = { {-2, -1}, {-1, -2}, {-2, 1}, {1, -2},
                                  {2, -1}, {-1, 2}, {2, 1}, {1, 2} };   /* (a, b) on (n1, l1) */
    /* literal tail of the table, lookup_tables.py:79-130 */
    static const char *extra[] = {
        "3041", "5041", "3241", "5241", "4130", "4150", "4132", "4152",
        "3948", "5948", "3748", "5748", "4839", "4859", "4837", "4857",
        "2002", "2042", "6042", "6082", "2402", "2442", "6442", "6482",
        "0220", "4220", "4260", "8260", "0224", "4224", "4264", "8264",
        "2907", "2947", "6947", "6987", "2507", "2547", "6547", "6587",
        "0729", "4729", "4769", "8769", "0725", "4725", "4765", "8765" };
    int n = 0, n1, l1, t, k;
    if (g_init_done) return;
    memset(g_label_of, 0xFF, sizeof(g_label_of));
    for (n1 = 0; n1 < 10; n1++)
        for (l1 = 0; l1 < 9; l1++) {
            for (t = 0; t < 9; t++)                      /* same row */
                if (t != l1) n = add_label(n, l1, n1, t, n1);
            for (t = 0; t < 10; t++)                     /* same column */
                if (t != n1) n = add_label(n, l1, n1, l1, t);
            for (k = 0; k < 8; k++) {                    /* knight jumps */
                int n2 = n1 + kn[k][0], l2 = l1 + kn[k][1];
                if (n2 >= 0 && n2 < 10 && l2 >= 0 && l2 < 9)
                    n = add_label(n, l1, n1, l2, n2);
            }
        }
    for (k = 0; k < (int)(sizeof(extra) / sizeof(extra[0])); k++)
        n = add_label(n, extra[k][0] - '0', extra[k][1] - '0', extra[k][2] - '0', extra[k][3] - '0');
    if (n != XQO_NLABELS) abort();
    g_init_done = 1;
}

int xqo_label_from(int label) { return g_from[label]; }
int xqo_label_to(int label) { return g_to[label]; }
int xqo_label_of(int from, int to) { return g_label_of[from][to]; }

void xqo_label_str(int label, char out[5])
{
    int f = g_from[label], t = g_to[label];
    out[0] = (char)('0' + f % 9); out[1] = (char)('0' + f / 9);
    out[2] = (char)('0' + t % 9); out[3] = (char)('0' + t / 9);
    out[4] = 0;
}

int xqo_label_parse(const char *mv)
{
    int x0 = mv[0] - '0', y0 = mv[1] - '0', x1 = mv[2] - '0', y1 = mv[3] - '0';
    int l;
    if (x0 < 0 || x0 > 8 || x1 < 0 || x1 > 8 || y0 < 0 || y0 > 9 || y1 < 0 || y1 > 9) return -1;
    l = g_label_of[SQ(x0, y0)][SQ(x1, y1)];
    return l == XQO_NOMOVE ? -1 : l;
}

/* flip_move, lookup_tables.py:50-56: (8-x0)(9-y0)(8-x1)(9-y1) == squares 89-s */
int xqo_flip_label(int label)
{
    return g_label_of[89 - g_from[label]][89 - g_to[label]];
}

/* ---- state string <-> board --------------------------------------------- */
/* state letters (common.py:32-64): r k(night) e m s(king) c p; UPPER = mover.
 * state_to_board (static_env.py:117-135): first row of the string is y = 9. */
static int8_t type_of_state_letter(char c)
{
    switch (c | 0x20) {
    case 'p': return XQ_PAWN;
    case 'c': return XQ_CANNON;
    case 'r': return XQ_ROOK;
    case 'k': return XQ_KNIGHT;
    case 'e': return XQ_ELEPHANT;
    case 'm': return XQ_ADVISOR;
    case 's': return XQ_KING;
    }
    return 0;
}

int xqo_state_to_board(const char *state, int8_t board[90])
{
    int x = 0, y = 9, k;
    memset(board, 0, 90);
    for (k = 0; state[k]; k++) {
        char ch = state[k];
        if (ch == ' ') break;
        if (ch == '/') { x = 0; y -= 1; }
        else if (ch >= '1' && ch <= '9') x += ch - '0';
        else {
            int8_t t = type_of_state_letter(ch);
            if (!t || x > 8 || y < 0) return -1;
            board[SQ(x, y)] = (ch >= 'A' && ch <= 'Z') ? t : (int8_t)-t;
            x += 1;
        }
    }
    return 0;
}

/* board_to_state, static_env.py:196-213 */
int xqo_board_to_state(const int8_t board[90], char *out)
{
    static const char L[8] = { '.', 'p', 'c', 'r', 'k', 'e', 'm', 's' };
    int i, j, c, n = 0;
    for (i = 9; i >= 0; i--) {
        c = 0;
        for (j = 0; j < 9; j++) {
            int8_t p = board[SQ(j, i)];
            if (p == 0) c++;
            else {
                if (c > 0) out[n++] = (char)('0' + c);
                out[n++] = p > 0 ? (char)(L[p] - 32) : L[-p];
                c = 0;
            }
        }
        if (c > 0) out[n++] = (char)('0' + c);
        if (i > 0) out[n++] = '/';
    }
    out[n] = 0;
    return n;
}

/* fliped_state, static_env.py:245-254: reverse rows, reverse each row, swap case */
void xqo_flip_board(const int8_t in[90], int8_t out[90])
{
    int s;
    for (s = 0; s < 90; s++) out[89 - s] = (int8_t)-in[s];
}

/* ---- move generation: static_env.py:256-348 ----------------------------- */
static int can_move(const int8_t *b, int x, int y)          /* :323-330 */
{
    if (x < 0 || x > 8) return 0;
    if (y < 0 || y > 9) return 0;
    return b[SQ(x, y)] <= 0;
}

static void x_board_from(const int8_t *b, int x, int y, int *l, int *r)   /* :332-339 */
{
    int ll = x - 1, rr = x + 1;
    while (ll > -1 && b[SQ(ll, y)] == 0) ll--;
    while (rr < 9 && b[SQ(rr, y)] == 0) rr++;
    *l = ll; *r = rr;
}

static void y_board_from(const int8_t *b, int x, int y, int *d, int *u)   /* :341-348 */
{
    int dd = y - 1, uu = y + 1;
    while (dd > -1 && b[SQ(x, dd)] == 0) dd--;
    while (uu < 10 && b[SQ(x, uu)] == 0) uu++;
    *d = dd; *u = uu;
}

/* mov_dir, common.py:66-76 (lower-case = mover entries) */
static const int DIR_K[4][2] = { {0, -1}, {1, 0}, {0, 1}, {-1, 0} };
static const int DIR_A[4][2] = { {-1, -1}, {1, -1}, {-1, 1}, {1, 1} };
static const int DIR_B[4][2] = { {-2, -2}, {2, -2}, {2, 2}, {-2, 2} };
static const int DIR_N[8][2] = { {-1, -2}, {1, -2}, {2, -1}, {2, 1}, {1, 2}, {-1, 2}, {-2, 1}, {-2, -1} };
static const int DIR_P[3][2] = { {0, 1}, {-1, 0}, {1, 0} };

#define EMIT(x0, y0, x1, y1) do { if (n < XQO_MAXMOVES) moves[n] = g_label_of[SQ(x0, y0)][SQ(x1, y1)]; n++; } while (0)

int xqo_legal_moves(const int8_t board[90], uint16_t moves[XQO_MAXMOVES])
{
    int n = 0, x, y, k;
    for (y = 0; y < 10; y++)
        for (x = 0; x < 9; x++) {
            int8_t ch = board[SQ(x, y)];
            const int (*dirs)[2] = 0;
            int nd = 0;
            if (ch <= 0) continue;
            switch (ch) {
            case XQ_KING: dirs = DIR_K; nd = 4; break;
            case XQ_ADVISOR: dirs = DIR_A; nd = 4; break;
            case XQ_ELEPHANT: dirs = DIR_B; nd = 4; break;
            case XQ_KNIGHT: dirs = DIR_N; nd = 8; break;
            case XQ_PAWN: dirs = DIR_P; nd = 3; break;
            }
            if (dirs) {
                for (k = 0; k < nd; k++) {
                    int dx = dirs[k][0], dy = dirs[k][1];
                    int x_ = x + dx, y_ = y + dy;
                    if (!can_move(board, x_, y_)) continue;
                    else if (ch == XQ_PAWN && y < 5 && x_ != x) continue;       /* :270 */
                    else if (ch == XQ_KNIGHT || ch == XQ_ELEPHANT) {            /* :272-276 */
                        /* int(d/2) truncates toward zero */
                        if (board[SQ(x + dx / 2, y + dy / 2)] != 0) continue;
                        else if (ch == XQ_ELEPHANT && y_ > 4) continue;
                    } else if (ch == XQ_KING || ch == XQ_ADVISOR) {             /* :277-281 */
                        if (x_ < 3 || x_ > 5) continue;
                        if (y_ > 2) continue;
                    }
                    EMIT(x, y, x_, y_);
                    if (ch == XQ_KING) {                                        /* :283-286 */
                        int d, u;
                        y_board_from(board, x, y, &d, &u);
                        if (u < 10 && board[SQ(x, u)] == -XQ_KING) EMIT(x, y, x, u);
                    }
                }
            } else if (ch == XQ_ROOK || ch == XQ_CANNON) {                      /* :288-320 */
                int l, r, d, u, t;
                x_board_from(board, x, y, &l, &r);
                y_board_from(board, x, y, &d, &u);
                for (t = l + 1; t < x; t++) EMIT(x, y, t, y);
                for (t = x + 1; t < r; t++) EMIT(x, y, t, y);
                for (t = d + 1; t < y; t++) EMIT(x, y, x, t);
                for (t = y + 1; t < u; t++) EMIT(x, y, x, t);
                if (ch == XQ_ROOK) {
                    if (can_move(board, l, y)) EMIT(x, y, l, y);
                    if (can_move(board, r, y)) EMIT(x, y, r, y);
                    if (can_move(board, x, d)) EMIT(x, y, x, d);
                    if (can_move(board, x, u)) EMIT(x, y, x, u);
                } else {
                    int l_, r_, d_, u_, dummy;
                    /* the screens may be off-board sentinels (-1 / 9 / 10); the
                     * reference then gets -2 / 10 / 11 which can_move rejects */
                    if (l >= 0) x_board_from(board, l, y, &l_, &dummy); else l_ = -2;
                    if (r <= 8) x_board_from(board, r, y, &dummy, &r_); else r_ = 10;
                    if (d >= 0) y_board_from(board, x, d, &d_, &dummy); else d_ = -2;
                    if (u <= 9) y_board_from(board, x, u, &dummy, &u_); else u_ = 11;
                    if (can_move(board, l_, y)) EMIT(x, y, l_, y);
                    if (can_move(board, r_, y)) EMIT(x, y, r_, y);
                    if (can_move(board, x, d_)) EMIT(x, y, x, d_);
                    if (can_move(board, x, u_)) EMIT(x, y, x, u_);
                }
            }
        }
    return n;
}

/* ---- done: static_env.py:14-77 ------------------------------------------ */
void xqo_done(const int8_t board[90], int need_check,
              int *over, int *v, int *final_move, int *check)
{
    int has_opp_king = 0, has_own_king = 0, s, i;
    int red_k[2] = { 0, 0 }, black_k[2] = { 0, 0 };
    int winner = 0;   /* 0 none, 1 red (mover), 2 black */
    *over = 0; *v = 0; *final_move = XQO_NOMOVE; *check = 0;
    for (s = 0; s < 90; s++) {
        if (board[s] == -XQ_KING) has_opp_king = 1;
        if (board[s] == XQ_KING) has_own_king = 1;
    }
    if (!has_opp_king) { *over = 1; *v = 1; return; }     /* 's' not in state :15-16 */
    if (!has_own_king) { *over = 1; *v = -1; return; }    /* 'S' not in state :17-18 */
    for (i = 0; i < 10; i++) {
        int j;
        for (j = 0; j < 9; j++) {
            if (board[SQ(j, i)] == XQ_KING) { red_k[0] = i; red_k[1] = j; }
            if (board[SQ(j, i)] == -XQ_KING) { black_k[0] = i; black_k[1] = j; }
        }
    }
    if (red_k[0] == 0 && red_k[1] == 0) { winner = 2; *v = -1; }          /* :33-35 */
    else if (black_k[0] == 0 && black_k[1] == 0) { winner = 1; *v = 1; }  /* :36-38 */
    else if (red_k[1] == black_k[1]) {                                     /* :39-49 */
        int has_block = 0;
        for (i = red_k[0] + 1; i < black_k[0]; i++)
            if (board[SQ(red_k[1], i)] != 0) { has_block = 1; break; }
        if (!has_block) { *v = 1; winner = 1; }
    }
    if (!winner) {                                                         /* :52-60 */
        uint16_t mv[XQO_MAXMOVES];
        int n = xqo_legal_moves(board, mv), k;
        int ks = SQ(black_k[1], black_k[0]);
        for (k = 0; k < n && k < XQO_MAXMOVES; k++)
            if (g_to[mv[k]] == ks) { winner = 1; *v = 1; *final_move = mv[k]; break; }
    }
    if (!winner && need_check) {                                           /* :61-73 */
        int8_t fb[90];
        uint16_t mv[XQO_MAXMOVES];
        int n, k, ks = 89 - SQ(red_k[1], red_k[0]);
        xqo_flip_board(board, fb);
        n = xqo_legal_moves(fb, mv);
        for (k = 0; k < n && k < XQO_MAXMOVES; k++)
            if (g_to[mv[k]] == ks) { *check = 1; break; }
    }
    *over = winner != 0;
}

/* ---- step / new_step: static_env.py:79-98 -------------------------------- */
int xqo_step(const int8_t board[90], int label, int8_t out[90], int *no_eat)
{
    int8_t tmp[90];
    int f = g_from[label], t = g_to[label];
    if (board[f] == 0) return -1;                /* ValueError, :81-82 */
    if (no_eat) *no_eat = board[t] == 0;
    memcpy(tmp, board, 90);
    tmp[t] = tmp[f];
    tmp[f] = 0;
    xqo_flip_board(tmp, out);
    return 0;
}

/* ---- planes: static_env.py:137-194 --------------------------------------- */
static void fill_planes(const int8_t board[90], float *planes)
{
    int s;
    for (s = 0; s < 90; s++) {
        int8_t p = board[s];
        if (p) {
            int ch = (p > 0 ? p - 1 : 7 + (-p) - 1);
            int x = s % 9, y = s / 9;
            planes[ch * 90 + (9 - y) * 9 + x] = 1.0f;   /* row i of the string is y = 9-i */
        }
    }
}

void xqo_planes(const int8_t board[90], float planes[14 * 90])
{
    memset(planes, 0, sizeof(float) * 14 * 90);
    fill_planes(board, planes);
}

void xqo_planes_hist(const int8_t board[90], const int8_t *prev, float planes[28 * 90])
{
    memset(planes, 0, sizeof(float) * 28 * 90);
    fill_planes(board, planes);
    if (prev) fill_planes(prev, planes + 14 * 90);
}

/* ---- perpetual check / chase: static_env.py:390-469 ---------------------- */
typedef struct { int n; uint32_t key[XQO_MAXMOVES]; } catch_set;

static void catch_add(catch_set *cs, uint32_t k)
{
    int i;
    for (i = 0; i < cs->n; i++) if (cs->key[i] == k) return;
    cs->key[cs->n++] = k;
}

/* get_catch_list, :423-454.  moves==NULL -> generate (also when the list is empty, :425-426). */
static void get_catch_list(const int8_t board[90], const uint16_t *moves, int nmoves, catch_set *out)
{
    uint16_t own[XQO_MAXMOVES];
    int k;
    out->n = 0;
    if (!moves || nmoves == 0) { nmoves = xqo_legal_moves(board, own); moves = own; }
    if (nmoves > XQO_MAXMOVES) nmoves = XQO_MAXMOVES;
    for (k = 0; k < nmoves; k++) {
        int8_t next[90];
        int no_eat = 1, f = g_from[moves[k]], t = g_to[moves[k]];
        xqo_step(board, moves[k], next, &no_eat);
        if (!no_eat) {
            uint16_t reply[XQO_MAXMOVES];
            int nr = xqo_legal_moves(next, reply), j, could_defend = 0;
            int dest = 89 - t;                       /* flip_move(mov)[2:] */
            if (nr > XQO_MAXMOVES) nr = XQO_MAXMOVES;
            for (j = 0; j < nr; j++) if (g_to[reply[j]] == dest) { could_defend = 1; break; }
            if (!could_defend) {
                int i = f / 9, m = t / 9;
                int8_t a = board[f], vct = board[t];
                if (a == XQ_PAWN && i <= 4) continue;            /* :443-444 */
                if (vct == -XQ_PAWN && m > 4) continue;          /* :447-448 */
                if (-vct == a) continue;                         /* same type = exchange :450-451 */
                catch_add(out, ((uint32_t)a << 24) | ((uint32_t)f << 16) | ((uint32_t)(-vct) << 8) | (uint32_t)t);
            }
        }
    }
}

int xqo_will_check_or_catch(const int8_t board[90], int label)
{
    int8_t state[90], black[90];
    uint16_t black_moves[XQO_MAXMOVES];
    int nb, k, s, ks = 0 /* red_k = [0,0] when no king, :397 */, target;
    catch_set first, second;
    if (xqo_step(board, label, state, 0) != 0) return -1;     /* ValueError */
    for (s = 0; s < 90; s++) if (state[s] == XQ_KING) ks = s; /* scan order i,j == s ascending, last wins */
    xqo_flip_board(state, black);
    nb = xqo_legal_moves(black, black_moves);
    if (nb > XQO_MAXMOVES) nb = XQO_MAXMOVES;
    target = 89 - ks;
    for (k = 0; k < nb; k++) if (g_to[black_moves[k]] == target) return 1;    /* :406-411 */
    get_catch_list(board, 0, 0, &first);                                      /* :413 */
    get_catch_list(black, black_moves, nb, &second);                          /* :414 */
    {
        int i, j, diff = 0;
        for (i = 0; i < second.n && !diff; i++) {
            int found = 0;
            for (j = 0; j < first.n; j++) if (first.key[j] == second.key[i]) { found = 1; break; }
            if (!found) diff = 1;
        }
        return diff && second.n >= first.n;                                   /* :415 */
    }
}

int xqo_be_catched(const int8_t board[90], int label)      /* :456-469 */
{
    int8_t black[90];
    uint16_t mv[XQO_MAXMOVES];
    int n, k, target = 89 - g_from[label];
    xqo_flip_board(board, black);
    n = xqo_legal_moves(black, mv);
    if (n > XQO_MAXMOVES) n = XQO_MAXMOVES;
    for (k = 0; k < n; k++) if (g_to[mv[k]] == target) return 1;
    return 0;
}

int xqo_has_attack_chessman(const int8_t board[90])        /* :471-479 */
{
    int s;
    for (s = 0; s < 90; s++) {
        int t = board[s] < 0 ? -board[s] : board[s];
        if (t == XQ_ROOK || t == XQ_KNIGHT || t == XQ_PAWN || t == XQ_CANNON) return 1;
    }
    return 0;
}

void xqo_batch_rules(const int8_t *boards, int n, uint16_t *moves, uint8_t *counts,
                     int8_t *over, int8_t *v, uint16_t *final_move, uint8_t *check, float *planes)
{
    int i;
    for (i = 0; i < n; i++) {
        const int8_t *b = boards + (size_t)i * 90;
        int o, vv, fm, ck, c;
        memset(moves + (size_t)i * XQO_MAXMOVES, 0xFF, sizeof(uint16_t) * XQO_MAXMOVES);
        c = xqo_legal_moves(b, moves + (size_t)i * XQO_MAXMOVES);
        counts[i] = (uint8_t)c;
        xqo_done(b, 1, &o, &vv, &fm, &ck);
        over[i] = (int8_t)o; v[i] = (int8_t)vv; final_move[i] = (uint16_t)fm; check[i] = (uint8_t)ck;
        if (planes) xqo_planes(b, planes + (size_t)i * 14 * 90);
    }
}
