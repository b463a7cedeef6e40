// tools/probes/simd_map_probe.hip -- a standalone check, NOT part of libczero.so: which SIMD of a CU the waves of a workgroup
// land on.  The residual-block kernels give their waves different roles (matrix / copy; the 192-filter block's six matrix
// waves on four SIMDs) and DESIGN.md argues about the balance of the SIMDs from "wave w -> SIMD w % 4"; this reads
// HW_REG_HW_ID (gfx9: WAVE_ID [3:0], SIMD_ID [5:4], CU_ID [11:8], SE_ID [15:13]) in every wave of workgroups shaped like the
// product kernels' (512 and 768 threads, most of the LDS so that a CU holds one workgroup) and prints the histogram of
// (wave index in the workgroup -> SIMD).
//     hipcc --offload-arch=gfx950 -O3 tools/probes/simd_map_probe.hip -o tools/probes/simd_map_probe && tools/probes/simd_map_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_probe(uint32_t* out, int spin)
{
    extern __shared__ unsigned char lds[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // HW_REG_HW_ID = 4, offset 0, size 32: simm16 = (size - 1) << 11 | offset << 6 | id
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if (spin) {                                   // keep the workgroup resident for a while (all CUs busy at once)
        const long long t0 = clock64();
        while (clock64() - t0 < spin) { }
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * nw + wave] = hw;
    if (threadIdx.x == 0) lds[0] = 1;
}

static void run(int threads, int lds_bytes, int blocks)
{
    const int nw = threads / 64;
    uint32_t* d;
    CK(hipMalloc(&d, sizeof(uint32_t) * blocks * nw));
    CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), lds_bytes, 0, d, 200000);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(blocks * nw);
    CK(hipMemcpy(h.data(), d, sizeof(uint32_t) * blocks * nw, hipMemcpyDeviceToHost));
    std::vector<int> hist(nw * 4, 0);
    int same_cu = 0;
    for (int b = 0; b < blocks; ++b) {
        bool one_cu = true;
        for (int w = 0; w < nw; ++w) {
            const uint32_t v = h[b * nw + w];
            hist[w * 4 + ((v >> 4) & 3)] += 1;
            if (((v >> 8) & 0xF) != ((h[b * nw] >> 8) & 0xF)) one_cu = false;
        }
        same_cu += one_cu;
    }
    printf("RESULT threads=%d lds=%d blocks=%d (workgroups on one CU: %d)\n", threads, lds_bytes, blocks, same_cu);
    for (int w = 0; w < nw; ++w)
        printf("RESULT   wave %2d -> SIMD0 %4d  SIMD1 %4d  SIMD2 %4d  SIMD3 %4d\n", w, hist[w * 4], hist[w * 4 + 1], hist[w * 4 + 2], hist[w * 4 + 3]);
    printf("RESULT   first workgroup, raw HW_ID per wave:");
    for (int w = 0; w < nw; ++w) printf(" %08x", h[w]);
    printf("\n");
    CK(hipFree(d));
}

int main()
{
    run(512, 156 * 1024, 256);        // k_resblock_c8: 4 matrix + 4 copy waves
    run(512, 152 * 1024, 256);        // k_resblock_ip_c8 (192 filters): 6 matrix + 2 copy waves
    run(768, 150 * 1024, 256);        // 12 waves
    run(512, 156 * 1024, 2048);       // several workgroups per CU in sequence
    return 0;
}
