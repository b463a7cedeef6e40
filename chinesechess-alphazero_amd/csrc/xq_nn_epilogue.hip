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

// ---- the two 1x1 head convolutions in one pass ----------------------------------------------------------------------
// Reference: policy head Conv2D(4, 1) -> BN -> ReLU -> Flatten and value head Conv2D(2, 1) -> BN -> ReLU -> Flatten
// (agent/model.py:56-57, 62-63; BatchNorm folded).  Both read the same trunk output, so ONE streaming pass over
// x[n][90][C] (channels-last) produces the NP + NV per-pixel outputs, written in the Flatten order of a
// channels-first tensor ([n][c][pixel]) that the dense layers expect.  HBM-bound: reads x once.
// 8 lanes per pixel; each lane takes every 8th float4 of the pixel's channel row (128 contiguous bytes per 8 lanes).
template <typename T, int NO>
__global__ __launch_bounds__(256) void k_head_convs(const T* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ pol,
                                                   float* __restrict__ val, int n_pixels, int channels, int np)
{
    extern __shared__ float ws[];                       // [NO][channels]
    for (int i = threadIdx.x; i < NO * channels; i += 256) ws[i] = w[i];
    __syncthreads();
    const int j = threadIdx.x & 7;
    const int nv = NO - np;
    for (long pix = (long)blockIdx.x * 32 + (threadIdx.x >> 3); pix < n_pixels; pix += (long)gridDim.x * 32) {
        float acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = 0.0f;
        const T* row = x + pix * channels;
        for (int c = j * 4; c < channels; c += 32) {
            float v[4];
            if (sizeof(T) == 4) {
                const float4 t = *reinterpret_cast<const float4*>(row + c);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                struct alignas(8) Q { T e[4]; };
                const Q t = *reinterpret_cast<const Q*>(row + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = to_f(t.e[k]);
            }
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const float4 wv = *reinterpret_cast<const float4*>(ws + o * channels + c);
                acc[o] += v[0] * wv.x + v[1] * wv.y + v[2] * wv.z + v[3] * wv.w;
            }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            acc[o] += __shfl_xor(acc[o], 1);
            acc[o] += __shfl_xor(acc[o], 2);
            acc[o] += __shfl_xor(acc[o], 4);
        }
        const long n = pix / 90;
        const int q = (int)(pix - n * 90);
#pragma unroll
        for (int o = 0; o < NO; ++o)
            if (j == o) {
                float r = acc[o] + bias[o];
                r = r > 0.0f ? r : 0.0f;
                if (o < np) pol[n * (np * 90) + o * 90 + q] = r;
                else val[n * (nv * 90) + (o - np) * 90 + q] = r;
            }
    }
}

template <typename T>
int launch_heads(const void* x, const float* w, const float* bias, float* pol, float* val, long n_pixels,
                 int channels, int np, hipStream_t st)
{
    long blocks = (n_pixels + 31) / 32;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL((k_head_convs<T, 6>), dim3((unsigned)blocks), dim3(256), 6 * channels * sizeof(float), st,
                       (const T*)x, w, bias, pol, val, (int)n_pixels, channels, np);
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

extern "C" int cz_head_convs(const void* x, int dtype, const float* w, const float* bias, float* policy_feat,
                             float* value_feat, int n_boards, int channels, int n_policy, int n_value, void* stream)
{
    if (!x || !w || !bias || !policy_feat || !value_feat || n_boards < 0 || channels <= 0 || channels % 32 != 0 ||
        n_policy + n_value != 6 || n_policy < 1 || n_value < 1 || channels > 1024) {
        czi_set_error("cz_head_convs: bad argument (channels % 32 == 0, n_policy + n_value == 6)");
        return CZ_ERR_ARG;
    }
    if (n_boards == 0) return CZ_OK;
    hipStream_t st = (hipStream_t)stream;
    const long n_pixels = (long)n_boards * 90;
    if (n_pixels > 0x7fffffffL) {
        czi_set_error("cz_head_convs: too many boards");
        return CZ_ERR_ARG;
    }
    int rc;
    switch (dtype) {
    case CZ_F32: rc = launch_heads<float>(x, w, bias, policy_feat, value_feat, n_pixels, channels, n_policy, st); break;
    case CZ_F16: rc = launch_heads<__half>(x, w, bias, policy_feat, value_feat, n_pixels, channels, n_policy, st); break;
    case CZ_BF16: rc = launch_heads<__hip_bfloat16>(x, w, bias, policy_feat, value_feat, n_pixels, channels, n_policy, st); break;
    default: czi_set_error("cz_head_convs: unknown dtype"); return CZ_ERR_ARG;
    }
    if (rc != CZ_OK) czi_set_error("cz_head_convs: launch failed");
    return rc;
}
