// xq_nn_epilogue.hip -- fused convolution epilogue for the policy/value ResNet trunk (gfx950).
//
// The trunk's convolutions run in MIOpen (MFMA implicit GEMM); PyTorch then applies the folded-BatchNorm bias,
// the residual add and the ReLU as three separate full passes over a [B*90, C] activation (1.5 GB each at the
// benchmark batch).  This kernel does  y = relu(x + bias[c] (+ residual))  in ONE pass, in place, 16 bytes per
// lane, channels-last (the channel is the fastest dimension, C % 8 == 0).  Pure HBM streaming: 2 (3 with the
// residual) x bytes(x) of traffic.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/czero.h"

extern "C" void czi_set_error(const char* msg);

namespace {

template <typename T> struct Vec;      // 16-byte vector of T
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<__half> { static constexpr int N = 8; };
template <> struct Vec<__hip_bfloat16> { static constexpr int N = 8; };

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
__device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
__device__ __forceinline__ void from_f(float& o, float v) { o = v; }
__device__ __forceinline__ void from_f(__half& o, float v) { o = __float2half(v); }
__device__ __forceinline__ void from_f(__hip_bfloat16& o, float v) { o = __float2bfloat16(v); }

template <typename T, bool RES>
__global__ __launch_bounds__(256) void k_bias_act(T* __restrict__ x, const T* __restrict__ bias,
                                                 const T* __restrict__ res, size_t nvec, int cvec, int relu)
{
    constexpr int N = Vec<T>::N;
    struct alignas(16) V { T e[N]; };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        V v = reinterpret_cast<V*>(x)[i];
        const V b = reinterpret_cast<const V*>(bias)[i % (size_t)cvec];
        V r;
        if (RES) r = reinterpret_cast<const V*>(res)[i];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            float f = to_f(v.e[k]) + to_f(b.e[k]);
            if (RES) f += to_f(r.e[k]);
            if (relu) f = f > 0.0f ? f : 0.0f;
            from_f(v.e[k], f);
        }
        reinterpret_cast<V*>(x)[i] = v;
    }
}

template <typename T>
int launch(void* x, const void* bias, const void* res, size_t n_elems, int channels, int relu, hipStream_t st)
{
    constexpr int N = Vec<T>::N;
    const size_t nvec = n_elems / N;
    const int cvec = channels / N;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (res)
        hipLaunchKernelGGL((k_bias_act<T, true>), dim3((unsigned)blocks), dim3(256), 0, st, (T*)x, (const T*)bias,
                           (const T*)res, nvec, cvec, relu);
    else
        hipLaunchKernelGGL((k_bias_act<T, false>), dim3((unsigned)blocks), dim3(256), 0, st, (T*)x, (const T*)bias,
                           (const T*)nullptr, nvec, cvec, relu);
    return hipGetLastError() == hipSuccess ? CZ_OK : CZ_ERR_HIP;
}

}  // namespace

extern "C" int cz_bias_act(void* x, const void* bias, const void* residual, size_t n_elems, int channels, int dtype,
                           int relu, void* stream)
{
    if (!x || !bias || channels <= 0 || channels % 8 != 0 || n_elems % (size_t)channels != 0) {
        czi_set_error("cz_bias_act: bad argument (channels must be a multiple of 8, x a whole number of rows)");
        return CZ_ERR_ARG;
    }
    if (n_elems == 0) return CZ_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (dtype) {
    case CZ_F32: rc = launch<float>(x, bias, residual, n_elems, channels, relu, st); break;
    case CZ_F16: rc = launch<__half>(x, bias, residual, n_elems, channels, relu, st); break;
    case CZ_BF16: rc = launch<__hip_bfloat16>(x, bias, residual, n_elems, channels, relu, st); break;
    default: czi_set_error("cz_bias_act: unknown dtype"); return CZ_ERR_ARG;
    }
    if (rc != CZ_OK) czi_set_error("cz_bias_act: launch failed");
    return rc;
}
