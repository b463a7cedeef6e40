// xq_kernels.hip -- batched Xiangqi rule kernels for gfx950 + their C-ABI entry points.
//
// One wavefront per board (workgroup = 64 threads), grid-stride over the batch.  These are
// the device replacements of the reference's per-position Python functions
// (cchess_alphazero/environment/static_env.py); the C-ABI is declared in include/czero.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "xq_rules.h"
#include "xq_tpb.h"
#include "../../include/czero.h"

using namespace xq;

namespace {

thread_local char g_err[256] = "";

int set_err(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return CZ_ERR_HIP;
}
int set_err_msg(int code, const char* what)
{
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

// batches at least this large run one board per LANE (k_rules_tpb); smaller ones one board per wavefront
constexpr int CZ_TPB_MIN_BOARDS = 256;

inline int grid_for(int n)
{
    // 256 CUs x up to 32 single-wave workgroups; the rest is grid-stride
    const int cap = 256 * 32;
    return n < cap ? (n > 0 ? n : 1) : cap;
}

#define CZ_LAUNCH_CHECK(name)                                        \
    do {                                                             \
        hipError_t e_ = hipGetLastError();                           \
        if (e_ != hipSuccess) return set_err(name, e_);              \
    } while (0)

// ---- kernels ------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_movegen(const int8_t* __restrict__ boards, int n,
                                               uint16_t* __restrict__ moves, uint8_t* __restrict__ counts)
{
    __shared__ RulesLDS w;
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        const int c = wave_movegen(w.bd[0], w.ml[0], w.plist);
        uint16_t* mo = moves + (size_t)i * MAXMOVES;
        mo[lane] = lane < c ? w.ml[0].lab[lane] : NOMOVE;
        mo[lane + 64] = lane + 64 < c ? w.ml[0].lab[lane + 64] : NOMOVE;
        if (lane == 0) counts[i] = (uint8_t)(c < 255 ? c : 255);
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void k_done(const int8_t* __restrict__ boards, int n, int need_check,
                                            int8_t* __restrict__ over, int8_t* __restrict__ v,
                                            uint16_t* __restrict__ final_move, uint8_t* __restrict__ check)
{
    __shared__ RulesLDS w;
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        const DoneResult r = wave_done(w.bd[0], w.bd[1], w.ml[0], w.ml[1], w.plist, need_check != 0);
        if (lane == 0) {
            over[i] = (int8_t)r.over;
            v[i] = (int8_t)r.v;
            final_move[i] = (uint16_t)r.final_move;
            if (check) check[i] = (uint8_t)r.check;
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void k_step(const int8_t* __restrict__ boards, const uint16_t* __restrict__ mv, int n,
                                            int8_t* __restrict__ out, uint8_t* __restrict__ no_eat)
{
    __shared__ RulesLDS w;
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        const int label = mv[i];
        int code = 0xFF;                               // 0xFF: invalid label or empty source (ValueError)
        if (label < NLABELS) {
            const int ft = label_ft(label);
            const int f = ft >> 8, t = ft & 0xFF;
            if (w.bd[0][f] != 0) {
                code = w.bd[0][t] == 0 ? 1 : 0;
                step_board(w.bd[0], f, t, w.bd[1]);
                store_board(w.bd[1], out + (size_t)i * NSQ);
            }
        }
        if (code == 0xFF) store_board(w.bd[0], out + (size_t)i * NSQ);
        if (lane == 0 && no_eat) no_eat[i] = (uint8_t)code;
        wave_sync();
    }
}

template <int DT>
__global__ __launch_bounds__(64) void k_encode(const int8_t* __restrict__ boards, int n, void* __restrict__ planes)
{
    __shared__ int8_t b[BOARD_LDS];
    constexpr size_t esz = DT == 0 ? 4 : (DT == 3 ? 1 : 2);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, b);
        wave_encode<DT>(b, (char*)planes + (size_t)i * 1260 * esz);
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void k_check_or_catch(const int8_t* __restrict__ boards, const uint16_t* __restrict__ mv,
                                                      int n, uint8_t* __restrict__ out)
{
    __shared__ RulesLDS w;
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        const int label = mv[i];
        int r = -1;
        if (label < NLABELS) r = wave_will_check_or_catch(w, w.bd[0], label);
        if (lane == 0) out[i] = (uint8_t)(r < 0 ? 0xFF : r);
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void k_be_catched(const int8_t* __restrict__ boards, const uint16_t* __restrict__ mv,
                                                  int n, uint8_t* __restrict__ out)
{
    __shared__ RulesLDS w;
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        const int label = mv[i];
        int r = 0xFF;
        if (label < NLABELS) r = wave_be_catched(w.bd[0], label_ft(label) >> 8, w.bd[1], w.ml[0], w.plist);
        if (lane == 0) out[i] = (uint8_t)r;
        wave_sync();
    }
}

__global__ __launch_bounds__(64) void k_has_attack(const int8_t* __restrict__ boards, int n, uint8_t* __restrict__ out)
{
    __shared__ int8_t b[BOARD_LDS];
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, b);
        const int r = wave_has_attack(b);
        if (lane == 0) out[i] = (uint8_t)r;
        wave_sync();
    }
}

// The micro-suite kernel of SURVEY 8(d): move-gen + done(need_check) + plane encode per board,
// 90 B in, 256 + 1 + 5 + planes out.
template <int DT>
__global__ __launch_bounds__(64) void k_rules_fused(const int8_t* __restrict__ boards, int n,
                                                   uint16_t* __restrict__ moves, uint8_t* __restrict__ counts,
                                                   int8_t* __restrict__ over, int8_t* __restrict__ v,
                                                   uint16_t* __restrict__ final_move, uint8_t* __restrict__ check,
                                                   void* __restrict__ planes)
{
    __shared__ RulesLDS w;
    constexpr size_t esz = DT == 0 ? 4 : (DT == 3 ? 1 : 2);
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        wave_encode<DT>(w.bd[0], (char*)planes + (size_t)i * 1260 * esz);
        DoneResult r = wave_done(w.bd[0], w.bd[1], w.ml[0], w.ml[1], w.plist, true);
        int c = r.nmoves;
        if (c < 0) c = wave_movegen(w.bd[0], w.ml[0], w.plist);       // early-decided positions still report their list
        uint16_t* mo = moves + (size_t)i * MAXMOVES;
        mo[lane] = lane < c ? w.ml[0].lab[lane] : NOMOVE;
        mo[lane + 64] = lane + 64 < c ? w.ml[0].lab[lane + 64] : NOMOVE;
        if (lane == 0) {
            counts[i] = (uint8_t)(c < 255 ? c : 255);
            over[i] = (int8_t)r.over;
            v[i] = (int8_t)r.v;
            final_move[i] = (uint16_t)r.final_move;
            check[i] = (uint8_t)r.check;
        }
        wave_sync();
    }
}

// ---- one board per lane (xq_tpb.h): the throughput form of move-gen + done + planes -------------------------
// 64 boards per wavefront.  Global traffic stays coalesced by staging through LDS: the 64 boards are loaded as
// one contiguous 5760-byte run, every lane then works on its own board and writes its ordered move list into its
// own LDS row, and the move lists / planes leave through wave-wide stores, one board at a time.
constexpr int TPB_BOARD_STRIDE = 100;    // bytes; 25 dwords: lanes reading the same square hit 64 different banks
constexpr int TPB_ROW_CAP = 64;          // moves a lane's LDS row holds; longer lists (rare: the mean is 40) are
                                         // redone by k_movegen_fix, one wavefront per board
constexpr int TPB_ROW_STRIDE = 66;       // uint16 per move-list row; 33 dwords: row starts fall on different banks

struct TpbLDS {
    int8_t bd[64 * TPB_BOARD_STRIDE];
    uint16_t rows[64 * TPB_ROW_STRIDE];
    uint8_t cnt[64];
};

// turn a board row (int8[90], y = 0 first) into plane codes in the planes' row order (string row i = y 9 - i):
// 0..13 = channel of the piece on that square, 0xFF = empty (static_env.py:137-156)
XQ_D void board_to_plane_codes(int8_t* b)
{
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 9; ++j) {
            const int a = i * 9 + j, c = (9 - i) * 9 + j;
            const int pa = b[a], pc = b[c];
            b[a] = (int8_t)(pc == 0 ? -1 : (pc > 0 ? pc - 1 : 6 - pc));
            b[c] = (int8_t)(pa == 0 ? -1 : (pa > 0 ? pa - 1 : 6 - pa));
        }
}

// which (channel, position) each of this lane's five 4-element chunks starts at: the same for every board
struct ChunkMap {
    uint16_t cpos[5];       // channel << 8 | position
    uint32_t tgt[5];        // the channel each of the chunk's four elements is compared with, one per byte (a chunk that runs past
                            // square 89 continues on square 0 of the next channel: the codes row repeats squares 0 .. 3 behind 89)
};
XQ_D ChunkMap make_chunk_map()
{
    ChunkMap m;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int o = (lane_id() + 64 * it) * 4;
        const int c = o / 90, pos = o - c * 90;
        m.cpos[it] = (uint16_t)((c << 8) | pos);
        uint32_t t = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) t |= (uint32_t)(pos + e < 90 ? c : c + 1) << (8 * e);
        m.tgt[it] = t;
    }
    return m;
}

#ifndef CZ_TPB_NT            // build switch: the planes of the one-board-per-lane kernel leave with nontemporal stores
#define CZ_TPB_NT 0
#endif
template <int DT>
XQ_D void tpb_write_planes(const int8_t* codes, void* __restrict__ out, const ChunkMap& cm)
{
    const int lane = lane_id();
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int q = lane + 64 * it;
        if (q >= 315) break;
        // the chunk's four codes with two aligned dword reads (round 6; four byte reads, a compare and the wrap test per element
        // before: this loop was half of the kernel's issue slots), compared with their channels bytewise
        const int pos = cm.cpos[it] & 0xFF;
        const uint32_t lo = *reinterpret_cast<const uint32_t*>(codes + (pos & ~3));
        const uint32_t hi = *reinterpret_cast<const uint32_t*>(codes + (pos & ~3) + 4);
        const uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(pos & 3)) ^ cm.tgt[it];
        int bit[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bit[e] = ((x >> (8 * e)) & 0xFFu) == 0u;
        if (DT == 0) {
            typedef __attribute__((ext_vector_type(4))) float f4;
            const f4 v = {(float)bit[0], (float)bit[1], (float)bit[2], (float)bit[3]};
            if (CZ_TPB_NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out) + q);     // (written once, read by another kernel much later)
            else reinterpret_cast<f4*>(out)[q] = v;
        } else if (DT == 1 || DT == 2) {
            const uint32_t one = (DT == 1) ? 0x3C00u : 0x3F80u;
            uint2 v;
            v.x = (bit[0] ? one : 0u) | ((bit[1] ? one : 0u) << 16);
            v.y = (bit[2] ? one : 0u) | ((bit[3] ? one : 0u) << 16);
            reinterpret_cast<uint2*>(out)[q] = v;
        } else {
            reinterpret_cast<uint32_t*>(out)[q] = (uint32_t)bit[0] | ((uint32_t)bit[1] << 8) | ((uint32_t)bit[2] << 16) |
                                                   ((uint32_t)bit[3] << 24);
        }
    }
}

// ---- (round 5) planes as zero-fill + scatter -----------------------------------------------------------------------------
// A board's 14 x 90 planes hold at most 32 ones.  The encoder above spends ~100 instructions per board on comparing every
// element's code (phase 3 was half of the kernel's issue slots); instead the block's planes -- one contiguous run of
// nb x 1260 elements -- are ZEROED with wide stores before the boards are even loaded (nothing to compute), and after the
// rules each lane sets the ones of ITS board: one store per piece.  The zero stores are acknowledged by the L2 (vmcnt = 0)
// before the first one is issued, so a one always lands on its zero.  MEASURED (profiles/r05_ab_tpb_scatter.log): with the
// zeros written at the start of the iteration the 1 M-board suite went from 1.55 to 2.51 ms -- by the time the ones arrive
// their lines have left the L2 and each becomes a partial write to HBM.  Build switch, default 0 (the encoder path).
#ifndef CZ_TPB_SCATTER
#define CZ_TPB_SCATTER 0
#endif
#ifndef CZ_TPB_ZERO_LATE
#define CZ_TPB_ZERO_LATE 1
#endif
template <int DT>
XQ_D void tpb_zero_planes(void* __restrict__ planes, int base, int nb)
{
    constexpr size_t esz = DT == 0 ? 4 : (DT == 3 ? 1 : 2);
    const int lane = lane_id();
    char* p = (char*)planes + (size_t)base * 1260 * esz;
    size_t bytes = (size_t)nb * 1260 * esz;                                  // a multiple of 4
    const int head = (int)((16u - (unsigned)((uintptr_t)p & 15u)) & 15u);    // bytes up to the first 16-byte boundary
    if (head) {
        if (lane * 4 < head) reinterpret_cast<uint32_t*>(p)[lane] = 0u;
        p += head;
        bytes -= (size_t)head;
    }
    uint4* p4 = reinterpret_cast<uint4*>(p);
    const int n16 = (int)(bytes >> 4);
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    int i = lane;
    for (; i + 192 < n16; i += 256) {                                        // four 1 KB rows per pass
        p4[i] = z; p4[i + 64] = z; p4[i + 128] = z; p4[i + 192] = z;
    }
    for (; i < n16; i += 64) p4[i] = z;
    const int tail = (int)(bytes & 15u) >> 2;
    if (lane < tail) reinterpret_cast<uint32_t*>(p4 + n16)[lane] = 0u;
}
// the ones of this lane's board (b: int8[90] in LDS, 4-byte aligned, y = 0 first): piece p on square (x, y) -> plane
// (p > 0 ? p - 1 : 6 - p), row 9 - y, column x (state_to_planes, static_env.py:137-156).  `ub` = the planes of the block's
// first board (wave-uniform), `lane_off` = this lane's board inside the block, in bytes.  The board is read once (23 words),
// the 90 squares are then walked in registers: no LDS wait per square.
template <int DT>
XQ_D void tpb_scatter_ones(const int8_t* b, char* __restrict__ ub, uint32_t lane_off)
{
    constexpr uint32_t esz = DT == 0 ? 4 : (DT == 3 ? 1 : 2);
    uint32_t w[23];
#pragma unroll
    for (int k = 0; k < 23; ++k) w[k] = reinterpret_cast<const uint32_t*>(b)[k];
#pragma unroll
    for (int s = 0; s < 90; ++s) {
        const int p = (int)(int8_t)(w[s >> 2] >> (8 * (s & 3)));
        if (p == 0) continue;
        const int y = s / 9, x = s - y * 9;
        // address = uniform base + (lane's board + plane) in a register + the square's position as the instruction's
        // immediate offset (lo is laundered so that the compiler does not keep 90 pre-added offsets in registers)
        uint32_t lo = lane_off;
        asm volatile("" : "+v"(lo));
        const uint32_t off = lo + (uint32_t)(p > 0 ? p - 1 : 6 - p) * (90u * esz);
        char* q = ub + (size_t)(((9 - y) * 9 + x) * (int)esz) + off;
        if (DT == 0) *reinterpret_cast<float*>(q) = 1.0f;
        else if (DT == 1) *reinterpret_cast<uint16_t*>(q) = (uint16_t)0x3C00u;
        else if (DT == 2) *reinterpret_cast<uint16_t*>(q) = (uint16_t)0x3F80u;
        else *reinterpret_cast<uint8_t*>(q) = (uint8_t)1;
    }
}

// outputs that are NULL are skipped (cz_movegen: moves + counts; cz_done: flags; cz_rules_fused: everything)
template <int DT>
__global__ __launch_bounds__(64) void k_rules_tpb(const int8_t* __restrict__ boards, int n, int need_check,
                                                 uint16_t* __restrict__ moves, uint8_t* __restrict__ counts,
                                                 int8_t* __restrict__ over, int8_t* __restrict__ v,
                                                 uint16_t* __restrict__ final_move, uint8_t* __restrict__ check,
                                                 void* __restrict__ planes)
{
    __shared__ TpbLDS L;
    constexpr size_t esz = DT == 0 ? 4 : (DT == 3 ? 1 : 2);
    const int lane = lane_id();
    const int nblk = (n + 63) / 64;
    const ChunkMap cm = make_chunk_map();
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int base = blk * 64;
        const int nb = n - base < 64 ? n - base : 64;
        // 0. the block's planes, zeroed (see tpb_zero_planes)
        if (CZ_TPB_SCATTER && !CZ_TPB_ZERO_LATE && planes) tpb_zero_planes<DT>(planes, base, nb);
        // 1. the boards of this block: one contiguous run of nb * 90 bytes, 2 bytes per lane per step
        const uint16_t* src = reinterpret_cast<const uint16_t*>(boards + (size_t)base * NSQ);
        for (int u = lane; u < nb * 45; u += 64) {
            const uint16_t w = src[u];
            const int k = u / 45, sq = (u - k * 45) * 2;
            L.bd[k * TPB_BOARD_STRIDE + sq] = (int8_t)(w & 0xFF);
            L.bd[k * TPB_BOARD_STRIDE + sq + 1] = (int8_t)(w >> 8);
        }
        __syncthreads();
        // 2. one board per lane
        if (lane < nb) {
            int8_t* b = L.bd + lane * TPB_BOARD_STRIDE;
            const TpbResult r = tpb_rules(b, L.rows + lane * TPB_ROW_STRIDE, need_check != 0, TPB_ROW_CAP);
            const int c = r.n < MAXMOVES ? r.n : MAXMOVES;
            L.cnt[lane] = (uint8_t)c;
            const int i = base + lane;
            if (counts) counts[i] = (uint8_t)(r.n < 255 ? r.n : 255);
            if (over) { over[i] = (int8_t)r.over; v[i] = (int8_t)r.v; final_move[i] = (uint16_t)r.final_move; }
            if (check) check[i] = (uint8_t)r.check;
            if (planes && !CZ_TPB_SCATTER) {
                board_to_plane_codes(b);
                *reinterpret_cast<uint32_t*>(b + 92) = 0xFFFFFFFFu;      // (bytes 94, 95 are read, never matched)
                b[90] = b[0]; b[91] = b[1]; b[92] = b[2]; b[93] = b[3];   // squares 0 .. 3 again behind square 89 (tpb_write_planes)
            }
        }
        if (CZ_TPB_SCATTER && planes) {
            // (ZERO_LATE: the zeros go out right before the ones, so that the ones still find their lines in the L2 -- zeroed
            //  at the start of the iteration they had been evicted by then and every one became a partial write to HBM)
            if (CZ_TPB_ZERO_LATE) tpb_zero_planes<DT>(planes, base, nb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the zeros have reached the L2
            if (lane < nb)
                tpb_scatter_ones<DT>(L.bd + lane * TPB_BOARD_STRIDE, (char*)planes + (size_t)base * 1260 * esz,
                                     (uint32_t)lane * (uint32_t)(1260 * esz));
        }
        __syncthreads();
        // 3. move lists and planes leave one board at a time, the whole wave storing contiguously
        for (int k = 0; k < nb; ++k) {
            if (moves && L.cnt[k] <= TPB_ROW_CAP) {          // longer lists: k_movegen_fix
                const int c = L.cnt[k];
                const uint32_t pair = lane < TPB_ROW_CAP / 2
                    ? *reinterpret_cast<const uint32_t*>(&L.rows[k * TPB_ROW_STRIDE + 2 * lane]) : 0u;
                const uint32_t lo = (2 * lane < c) ? (pair & 0xFFFFu) : (uint32_t)NOMOVE;
                const uint32_t hi = (2 * lane + 1 < c) ? (pair >> 16) : (uint32_t)NOMOVE;
                reinterpret_cast<uint32_t*>(moves + (size_t)(base + k) * MAXMOVES)[lane] = lo | (hi << 16);
            }
            if (!CZ_TPB_SCATTER && planes)
                tpb_write_planes<DT>(L.bd + k * TPB_BOARD_STRIDE, (char*)planes + (size_t)(base + k) * 1260 * esz, cm);
        }
        __syncthreads();
    }
}

// move lists longer than a row of k_rules_tpb: recomputed here, one wavefront per such board
__global__ __launch_bounds__(64) void k_movegen_fix(const int8_t* __restrict__ boards, int n,
                                                   uint16_t* __restrict__ moves, const uint8_t* __restrict__ counts)
{
    __shared__ RulesLDS w;
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        if (counts[i] <= TPB_ROW_CAP) continue;
        load_board(boards + (size_t)i * NSQ, w.bd[0]);
        const int c = wave_movegen(w.bd[0], w.ml[0], w.plist);
        uint16_t* mo = moves + (size_t)i * MAXMOVES;
        mo[lane] = lane < c ? w.ml[0].lab[lane] : NOMOVE;
        mo[lane + 64] = lane + 64 < c ? w.ml[0].lab[lane + 64] : NOMOVE;
        wave_sync();
    }
}

inline int grid_for_tpb(int n)
{
    const int nblk = (n + 63) / 64;
    // single-wave workgroups per CU: 14.9 KB of LDS and 146 VGPRs each allow 10; CZ_TPB_BLOCKS_PER_CU overrides (tuning)
    static const int per_cu = [] {
        const char* e = getenv("CZ_TPB_BLOCKS_PER_CU");
        const int v = e ? atoi(e) : 0;
        return v > 0 && v <= 64 ? v : 10;   // measured: 2.82 TB/s at 10, 2.53 at 8 and 16, 2.23 at 12
    }();
    const int cap = 256 * per_cu;
    return nblk < cap ? nblk : cap;
}

}  // namespace

// ---- C-ABI ----------------------------------------------------------------------------
extern "C" {

void czi_set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

int cz_version(void) { return CZ_VERSION; }

const char* cz_last_error(void) { return g_err; }

int cz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cz_label_tables(uint16_t* label_of, uint16_t* lab_ft)
{
    if (label_of) memcpy(label_of, h_tab.label_of, sizeof(uint16_t) * NSQ * NSQ);
    if (lab_ft) memcpy(lab_ft, h_tab.lab_ft, sizeof(uint16_t) * NLABELS);
    return CZ_OK;
}

int cz_movegen(const int8_t* boards, int n, uint16_t* moves, uint8_t* counts, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !moves || !counts) return set_err_msg(CZ_ERR_ARG, "cz_movegen: bad argument");
    if (n >= CZ_TPB_MIN_BOARDS) {
        hipLaunchKernelGGL(k_rules_tpb<0>, dim3(grid_for_tpb(n)), dim3(64), 0, (hipStream_t)stream, boards, n, 0, moves,
                           counts, (int8_t*)nullptr, (int8_t*)nullptr, (uint16_t*)nullptr, (uint8_t*)nullptr,
                           (void*)nullptr);
        hipLaunchKernelGGL(k_movegen_fix, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, n, moves, counts);
        CZ_LAUNCH_CHECK("cz_movegen");
        return CZ_OK;
    }
    hipLaunchKernelGGL(k_movegen, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, n, moves, counts);
    CZ_LAUNCH_CHECK("cz_movegen");
    return CZ_OK;
}

int cz_done(const int8_t* boards, int n, int need_check, int8_t* over, int8_t* v, uint16_t* final_move,
            uint8_t* check, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !over || !v || !final_move || (need_check && !check))
        return set_err_msg(CZ_ERR_ARG, "cz_done: bad argument");
    if (n >= CZ_TPB_MIN_BOARDS) {
        hipLaunchKernelGGL(k_rules_tpb<0>, dim3(grid_for_tpb(n)), dim3(64), 0, (hipStream_t)stream, boards, n, need_check,
                           (uint16_t*)nullptr, (uint8_t*)nullptr, over, v, final_move, need_check ? check : (uint8_t*)nullptr,
                           (void*)nullptr);
        CZ_LAUNCH_CHECK("cz_done");
        return CZ_OK;
    }
    hipLaunchKernelGGL(k_done, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, n, need_check, over, v,
                       final_move, check);
    CZ_LAUNCH_CHECK("cz_done");
    return CZ_OK;
}

int cz_step(const int8_t* boards, const uint16_t* moves, int n, int8_t* out, uint8_t* no_eat, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !moves || !out) return set_err_msg(CZ_ERR_ARG, "cz_step: bad argument");
    hipLaunchKernelGGL(k_step, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, moves, n, out, no_eat);
    CZ_LAUNCH_CHECK("cz_step");
    return CZ_OK;
}

int cz_encode(const int8_t* boards, int n, void* planes, int dtype, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !planes) return set_err_msg(CZ_ERR_ARG, "cz_encode: bad argument");
    const dim3 g(grid_for(n)), b(64);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
    case CZ_F32: hipLaunchKernelGGL(k_encode<0>, g, b, 0, s, boards, n, planes); break;
    case CZ_F16: hipLaunchKernelGGL(k_encode<1>, g, b, 0, s, boards, n, planes); break;
    case CZ_BF16: hipLaunchKernelGGL(k_encode<2>, g, b, 0, s, boards, n, planes); break;
    case CZ_U8: hipLaunchKernelGGL(k_encode<3>, g, b, 0, s, boards, n, planes); break;
    default: return set_err_msg(CZ_ERR_ARG, "cz_encode: unknown dtype");
    }
    CZ_LAUNCH_CHECK("cz_encode");
    return CZ_OK;
}

int cz_check_or_catch(const int8_t* boards, const uint16_t* moves, int n, uint8_t* out, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !moves || !out) return set_err_msg(CZ_ERR_ARG, "cz_check_or_catch: bad argument");
    hipLaunchKernelGGL(k_check_or_catch, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, moves, n, out);
    CZ_LAUNCH_CHECK("cz_check_or_catch");
    return CZ_OK;
}

int cz_be_catched(const int8_t* boards, const uint16_t* moves, int n, uint8_t* out, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !moves || !out) return set_err_msg(CZ_ERR_ARG, "cz_be_catched: bad argument");
    hipLaunchKernelGGL(k_be_catched, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, moves, n, out);
    CZ_LAUNCH_CHECK("cz_be_catched");
    return CZ_OK;
}

int cz_has_attack(const int8_t* boards, int n, uint8_t* out, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !out) return set_err_msg(CZ_ERR_ARG, "cz_has_attack: bad argument");
    hipLaunchKernelGGL(k_has_attack, dim3(grid_for(n)), dim3(64), 0, (hipStream_t)stream, boards, n, out);
    CZ_LAUNCH_CHECK("cz_has_attack");
    return CZ_OK;
}

int cz_rules_fused(const int8_t* boards, int n, uint16_t* moves, uint8_t* counts, int8_t* over, int8_t* v,
                   uint16_t* final_move, uint8_t* check, void* planes, int dtype, void* stream)
{
    if (n == 0) return CZ_OK;
    if (n < 0 || !boards || !moves || !counts || !over || !v || !final_move || !check || !planes)
        return set_err_msg(CZ_ERR_ARG, "cz_rules_fused: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (n >= CZ_TPB_MIN_BOARDS) {
        const dim3 g(grid_for_tpb(n)), b(64);
        switch (dtype) {
        case CZ_F32: hipLaunchKernelGGL(k_rules_tpb<0>, g, b, 0, s, boards, n, 1, moves, counts, over, v, final_move, check, planes); break;
        case CZ_F16: hipLaunchKernelGGL(k_rules_tpb<1>, g, b, 0, s, boards, n, 1, moves, counts, over, v, final_move, check, planes); break;
        case CZ_BF16: hipLaunchKernelGGL(k_rules_tpb<2>, g, b, 0, s, boards, n, 1, moves, counts, over, v, final_move, check, planes); break;
        case CZ_U8: hipLaunchKernelGGL(k_rules_tpb<3>, g, b, 0, s, boards, n, 1, moves, counts, over, v, final_move, check, planes); break;
        default: return set_err_msg(CZ_ERR_ARG, "cz_rules_fused: unknown dtype");
        }
        hipLaunchKernelGGL(k_movegen_fix, dim3(grid_for(n)), dim3(64), 0, s, boards, n, moves, counts);
        CZ_LAUNCH_CHECK("cz_rules_fused");
        return CZ_OK;
    }
    const dim3 g(grid_for(n)), b(64);
    switch (dtype) {
    case CZ_F32: hipLaunchKernelGGL(k_rules_fused<0>, g, b, 0, s, boards, n, moves, counts, over, v, final_move, check, planes); break;
    case CZ_F16: hipLaunchKernelGGL(k_rules_fused<1>, g, b, 0, s, boards, n, moves, counts, over, v, final_move, check, planes); break;
    case CZ_BF16: hipLaunchKernelGGL(k_rules_fused<2>, g, b, 0, s, boards, n, moves, counts, over, v, final_move, check, planes); break;
    case CZ_U8: hipLaunchKernelGGL(k_rules_fused<3>, g, b, 0, s, boards, n, moves, counts, over, v, final_move, check, planes); break;
    default: return set_err_msg(CZ_ERR_ARG, "cz_rules_fused: unknown dtype");
    }
    CZ_LAUNCH_CHECK("cz_rules_fused");
    return CZ_OK;
}

}  // extern "C"
