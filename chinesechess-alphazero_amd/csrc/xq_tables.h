// xq_tables.h -- compile-time tables for the Xiangqi engine (gfx950 build and host side).
//
// The 2086-entry policy label set of the reference (create_action_labels,
// cchess_alphazero/environment/lookup_tables.py:62-132) is rebuilt here at compile time:
//   label_of[from*90+to] -> label index (0xFFFF = not a label)
//   lab_ft[label]        -> (from << 8) | to
// Squares are s = y*9 + x with y = 0 the mover's back rank (static_env.py:117-135).
#pragma once
#include <stdint.h>

namespace xq {

constexpr int NSQ = 90;
constexpr int NLABELS = 2086;
constexpr int MAXMOVES = 128;          // capacity of one ordered move list (max pseudo-legal = 123 + duplicates)
constexpr uint16_t NOMOVE = 0xFFFF;

// piece codes on the int8 board: +t mover, -t opponent (order of Fen_2_Idx, lookup_tables.py:27-42)
enum : int { EMPTY = 0, PAWN = 1, CANNON = 2, ROOK = 3, KNIGHT = 4, ELEPHANT = 5, ADVISOR = 6, KING = 7 };

struct Tables {
    uint16_t label_of[NSQ * NSQ];
    uint16_t lab_ft[NLABELS + 2];
    uint16_t base[NSQ];         // first label of each source square's block (8 row + 9 file + knight moves)
    uint8_t kvalid[NSQ];        // which of the 8 knight offsets (label order) stay on the board
};

constexpr Tables make_tables()
{
    Tables t{};
    for (int i = 0; i < NSQ * NSQ; ++i) t.label_of[i] = NOMOVE;
    for (int i = 0; i < NLABELS + 2; ++i) t.lab_ft[i] = 0;
    int n = 0;
    auto add = [&](int x0, int y0, int x1, int y1) {
        const int f = y0 * 9 + x0, d = y1 * 9 + x1;
        t.label_of[f * NSQ + d] = (uint16_t)n;
        t.lab_ft[n] = (uint16_t)((f << 8) | d);
        ++n;
    };
    // (a, b) applied to (row n1, col l1), in the reference's order
    const int kn[8][2] = {{-2, -1}, {-1, -2}, {-2, 1}, {1, -2}, {2, -1}, {-1, 2}, {2, 1}, {1, 2}};
    for (int n1 = 0; n1 < 10; ++n1)
        for (int l1 = 0; l1 < 9; ++l1) {
            t.base[n1 * 9 + l1] = (uint16_t)n;
            t.kvalid[n1 * 9 + l1] = 0;
            for (int k = 0; k < 8; ++k) {
                const int n2 = n1 + kn[k][0], l2 = l1 + kn[k][1];
                if (n2 >= 0 && n2 < 10 && l2 >= 0 && l2 < 9) t.kvalid[n1 * 9 + l1] |= (uint8_t)(1u << k);
            }
            for (int c = 0; c < 9; ++c)
                if (c != l1) add(l1, n1, c, n1);
            for (int r = 0; r < 10; ++r)
                if (r != n1) add(l1, n1, l1, r);
            for (int k = 0; k < 8; ++k) {
                const int n2 = n1 + kn[k][0], l2 = l1 + kn[k][1];
                if (n2 >= 0 && n2 < 10 && l2 >= 0 && l2 < 9) add(l1, n1, l2, n2);
            }
        }
    // advisor and elephant moves appended literally (lookup_tables.py:79-130): x0 y0 x1 y1
    const int extra[48][4] = {
        {3, 0, 4, 1}, {5, 0, 4, 1}, {3, 2, 4, 1}, {5, 2, 4, 1}, {4, 1, 3, 0}, {4, 1, 5, 0}, {4, 1, 3, 2}, {4, 1, 5, 2},
        {3, 9, 4, 8}, {5, 9, 4, 8}, {3, 7, 4, 8}, {5, 7, 4, 8}, {4, 8, 3, 9}, {4, 8, 5, 9}, {4, 8, 3, 7}, {4, 8, 5, 7},
        {2, 0, 0, 2}, {2, 0, 4, 2}, {6, 0, 4, 2}, {6, 0, 8, 2}, {2, 4, 0, 2}, {2, 4, 4, 2}, {6, 4, 4, 2}, {6, 4, 8, 2},
        {0, 2, 2, 0}, {4, 2, 2, 0}, {4, 2, 6, 0}, {8, 2, 6, 0}, {0, 2, 2, 4}, {4, 2, 2, 4}, {4, 2, 6, 4}, {8, 2, 6, 4},
        {2, 9, 0, 7}, {2, 9, 4, 7}, {6, 9, 4, 7}, {6, 9, 8, 7}, {2, 5, 0, 7}, {2, 5, 4, 7}, {6, 5, 4, 7}, {6, 5, 8, 7},
        {0, 7, 2, 9}, {4, 7, 2, 9}, {4, 7, 6, 9}, {8, 7, 6, 9}, {0, 7, 2, 5}, {4, 7, 2, 5}, {4, 7, 6, 5}, {8, 7, 6, 5}};
    for (int k = 0; k < 48; ++k) add(extra[k][0], extra[k][1], extra[k][2], extra[k][3]);
    // n must equal NLABELS; checked by static_assert on label count below
    t.lab_ft[NLABELS] = (uint16_t)n;       // sentinel slot carries the count for the static_assert
    return t;
}

}  // namespace xq
