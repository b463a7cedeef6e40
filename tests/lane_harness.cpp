// tests/lane_harness.cpp -- CPU harness for the per-lane rule functions of the HIP engine
// (chinesechess-alphazero_amd/csrc/xq_lane.h).  Built with g++ by tests/test_lane_cpu.py; it
// emulates the wave-level ordered compaction with plain loops so the per-square logic and the
// constexpr tables can be checked against the oracle without a GPU.
#include <stdint.h>
#include <string.h>
#include "../chinesechess-alphazero_amd/csrc/xq_lane.h"
#include "../chinesechess-alphazero_amd/csrc/xq_tpb.h"

using namespace xq;

extern "C" {

int lane_movegen(const int8_t* board, uint16_t* lab, uint16_t* ft)
{
    // what the wave does with ballots: three 90-bit square sets + the mover's pieces in square order
    Set90 occ{0, 0}, own{0, 0}, oking{0, 0};
    auto set = [](Set90& m, int s) { if (s < 64) m.lo |= 1ull << s; else m.hi |= 1ull << (s - 64); };
    for (int s = 0; s < NSQ; ++s) {
        if (board[s] != 0) set(occ, s);
        if (board[s] > 0) set(own, s);
        if (board[s] == -KING) set(oking, s);
    }
    int off = 0;
    for (int s = 0; s < NSQ; ++s) {           // exclusive prefix sum over the pieces in square order
        if (board[s] <= 0) continue;
        const int c = gen_piece<false>(board[s], s, occ, own, oking, nullptr, nullptr, 0);
        if (c) gen_piece<true>(board[s], s, occ, own, oking, lab, ft, off);
        off += c;
    }
    return off;
}

// the quad-of-lanes generator (quad_plan / quad_emit: what wave_movegen runs); the cross-lane sums of the wave emulated
// with loops; formula: labels by arithmetic
int lane_movegen_quad(const int8_t* board, uint16_t* lab, uint16_t* ft, int formula)
{
    Set90 occ{0, 0}, own{0, 0}, oking{0, 0};
    auto set = [](Set90& m, int s) { if (s < 64) m.lo |= 1ull << s; else m.hi |= 1ull << (s - 64); };
    for (int s = 0; s < NSQ; ++s) {
        if (board[s] != 0) set(occ, s);
        if (board[s] > 0) set(own, s);
        if (board[s] == -KING) set(oking, s);
    }
    int total = 0;
    for (int s = 0; s < NSQ; ++s) {
        if (board[s] <= 0) continue;
        QuadPlan pl[4];
        int sum_a = 0, sum_b = 0;
        for (int q = 0; q < 4; ++q) { pl[q] = quad_plan(board[s], s, q, occ, own, oking); sum_a += pl[q].n_a; sum_b += pl[q].n_b; }
        if (sum_a + sum_b != gen_piece<false>(board[s], s, occ, own, oking, nullptr, nullptr, 0)) return -1;
        int off_a = 0, off_b = 0;
        for (int q = 0; q < 4; ++q) {
            if (pl[q].n_a | pl[q].n_b)
                quad_emit(board[s], s, q, pl[q], lab, ft, total + off_a, total + sum_a + off_b, formula != 0, 512);
            off_a += pl[q].n_a; off_b += pl[q].n_b;
        }
        total += sum_a + sum_b;
    }
    return total;
}

// file_bits (multiply-gather) against the definition, on pseudo-random 90-bit sets
int lane_file_bits_mismatches(uint64_t seed, int n)
{
    int bad = 0;
    uint64_t z = seed;
    auto next = [&]() { z += 0x9E3779B97F4A7C15ULL; return mix64(z); };
    for (int i = 0; i < n; ++i) {
        Set90 m{next(), next() & ((1ull << 26) - 1ull)};
        if (i % 7 == 0) m.lo = ~0ull;
        if (i % 11 == 0) m.hi = (1ull << 26) - 1ull;
        for (int x = 0; x < 9; ++x) {
            uint32_t want = 0;
            for (int y = 0; y < 10; ++y) want |= (uint32_t)has(m, 9 * y + x) << y;
            if (file_bits(m, x) != want) ++bad;
        }
    }
    return bad;
}

void lane_planes(const int8_t* board, float* planes)
{
    for (int o = 0; o < 1260; ++o) planes[o] = (float)plane_bit(board, o);
}

void lane_tables(uint16_t* label_of_out, uint16_t* lab_ft_out)
{
    memcpy(label_of_out, h_tab.label_of, sizeof(uint16_t) * NSQ * NSQ);
    memcpy(lab_ft_out, h_tab.lab_ft, sizeof(uint16_t) * NLABELS);
}

// every row / file / knight label must be reproduced by the table-free formula
int lane_label_formula_mismatches(void)
{
    int bad = 0;
    for (int l = 0; l < NLABELS; ++l) {
        const int f = h_tab.lab_ft[l] >> 8, t = h_tab.lab_ft[l] & 0xFF;
        const int dx = t % 9 - f % 9, dy = t / 9 - f / 9;
        const bool line = dx == 0 || dy == 0;
        const bool knight = (dx * dx + dy * dy) == 5;
        if ((line || knight) && label_of_line_or_knight(f, t) != l) ++bad;
        if (line || knight) {                          // the per-piece form the move generator uses
            int base; uint32_t valid;
            label_block(f, &base, &valid);
            if (label_in_block(base, valid, f, t) != l) ++bad;
        }
    }
    // advisor / elephant rows: step k of the piece from every square against the (from, to) table
    for (int e = 0; e < 2; ++e)
        for (int sq = 0; sq < NSQ; ++sq) {
            const uint64_t row = ae_label_row(e, sq);
            const uint64_t codes = step_row(e ? ELEPHANT : ADVISOR);
            for (int k = 0; k < 4; ++k) {
                const int code = (int)((codes >> (8 * k)) & 0xFF);
                const int x_ = sq % 9 + (code & 7) - 2, y_ = sq / 9 + (code >> 3) - 2;
                const uint16_t want = (x_ < 0 || x_ > 8 || y_ < 0 || y_ > 9) ? (uint16_t)NOMOVE : label_of(sq, y_ * 9 + x_);
                if ((uint16_t)((row >> (16 * k)) & 0xFFFF) != want) ++bad;
            }
        }
    for (int p = 0; p < 8; ++p) {                      // immediates against the step table
        if (step_count_imm(p) != step_count(p)) ++bad;
        for (int k = 0; k < step_count(p); ++k)
            if ((int)((step_row(p) >> (8 * k)) & 0xFF) != step_code(p, k)) ++bad;
    }
    return bad;
}

// the thread-per-board rules (xq_tpb.h), exactly as a GPU lane runs them
int tpb_board(const int8_t* board, int need_check, uint16_t* lab, int* out /* over, v, final_move, check */)
{
    const TpbResult r = tpb_rules(board, lab, need_check != 0);
    out[0] = r.over; out[1] = r.v; out[2] = r.final_move; out[3] = r.check;
    return r.n;
}

int lane_nibble_roundtrip(void)
{
    for (int p = -7; p <= 7; ++p)
        if (piece_of_nib(nib_of(p)) != p || nib_of(p) > 15) return 0;
    return 1;
}

}
