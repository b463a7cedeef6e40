// tools/probes/hbm_counter_probe.hip -- a standalone calibration, NOT part of libczero.so (round 6):
// what rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count in the access patterns of the tower kernels' copy
// waves.  MI355X_MICROARCH.md (HBM): FETCH_SIZE counts a wide coalesced read at half its bytes; "other access widths and
// WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
//   k_read16        16 B per lane, consecutive lanes (tile_load / the chain's entry loads)
//   k_write16       16 B per lane, consecutive lanes (store_tile's f16 pass)
//   k_write_item    the chain's exit (k_tower_c6 convert(), board >= 0): a thread = one (pixel, 32-channel block) item:
//                   4 x 16 B consecutive (its 64 B of the f16 row) + two bf6 pieces of 16 B head + 8 B tail at a 32 B pitch
//                   (24 of every 32 bytes of the c6 row written; 8 B holes)
//   k_write_piece   store_tile_c6's pass B alone: only the two 24-byte pieces per item
//   k_write4        4 B per lane, consecutive lanes (the head features: HEADS exits)
// Each kernel touches `bytes` (printed) exactly once; run under
//     rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- tools/probes/hbm_counter_probe     (and again with --pmc FETCH_SIZE)
// and divide the counter (KiB) by the printed figure: tools/summarize_hbm_probe.py.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_counter_probe.hip -o tools/probes/hbm_counter_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_read16(const uint4* __restrict__ src, size_t n16, uint4* __restrict__ sink)
{
    uint4 a = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w;
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) sink[threadIdx.x] = a;       // (never true for the zero-filled source)
}

__global__ __launch_bounds__(256) void k_write16(uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void k_write4(float* __restrict__ dst, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        dst[i] = (float)i;
}

// boards of 90 pixel rows: f16 row 256 B (yh), c6 row 256 B (yc); item i of a board = (pixel i >> 2, block i & 3)
__device__ __forceinline__ int c6_chunk(int kind, int blk32) { return 8 * kind + 4 * (blk32 >> 1) + 2 * (blk32 & 1); }

template <bool F16_TOO>
__global__ __launch_bounds__(256) void k_write_item(unsigned char* __restrict__ yh, unsigned char* __restrict__ yc, int n_boards)
{
    for (int board = blockIdx.x; board < n_boards; board += gridDim.x) {
        const size_t ebase = (size_t)board * 90 * 256;
        for (int i = threadIdx.x; i < 360; i += 256) {
            const int qq = i >> 2, blk = i & 3;
            if (F16_TOO) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    reinterpret_cast<uint4*>(yh + ebase)[qq * 16 + blk * 4 + k] = make_uint4(i, k, 2u, 3u);
            }
            unsigned char* row = yc + ebase + (size_t)qq * 256;
            *reinterpret_cast<uint4*>(row + 16 * c6_chunk(0, blk)) = make_uint4(i, 1u, 2u, 3u);
            *reinterpret_cast<uint2*>(row + 16 * c6_chunk(0, blk) + 16) = make_uint2(i, 1u);
            *reinterpret_cast<uint4*>(row + 16 * c6_chunk(1, blk)) = make_uint4(i, 1u, 2u, 3u);
            *reinterpret_cast<uint2*>(row + 16 * c6_chunk(1, blk) + 16) = make_uint2(i, 1u);
        }
    }
}

int main(int argc, char** argv)
{
    const int n_boards = argc > 1 ? atoi(argv[1]) : 32768;           // one queue's worth of boards
    const size_t img = (size_t)n_boards * 90 * 256;                   // bytes of an f16 array = of a c6 image (755 MB at 32768)
    unsigned char *a, *b;
    CK(hipMalloc(&a, img));
    CK(hipMalloc(&b, img));
    CK(hipMemset(a, 0, img));
    CK(hipMemset(b, 0, img));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        k_read16<<<grid, 256>>>(reinterpret_cast<const uint4*>(a), img / 16, reinterpret_cast<uint4*>(b));
        k_write16<<<grid, 256>>>(reinterpret_cast<uint4*>(a), img / 16);
        k_write4<<<grid, 256>>>(reinterpret_cast<float*>(a), img / 4);
        k_write_item<true><<<256, 256>>>(a, b, n_boards);
        k_write_item<false><<<256, 256>>>(a, b, n_boards);
        CK(hipDeviceSynchronize());
    }
    // bytes each kernel touches (exactly once)
    printf("{\"boards\": %d, \"k_read16\": {\"read\": %zu}, \"k_write16\": {\"write\": %zu}, \"k_write4\": {\"write\": %zu}, "
           "\"k_write_item<true>\": {\"write\": %zu, \"span\": %zu}, \"k_write_item<false>\": {\"write\": %zu, \"span\": %zu}}\n",
           n_boards, img, img, img, img + (size_t)n_boards * 360 * 48, 2 * img, (size_t)n_boards * 360 * 48, img);
    return 0;
}
