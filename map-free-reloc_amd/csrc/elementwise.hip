// elementwise.hip -- fused epilogues for the SuperPoint conv stack (NCHW fp32) on gfx950.
//
// Upstream SuperPoint (un-vendored; called from matchers.py:93-120) applies
// relu(conv + bias) after every 3x3 convolution and a 2x2 max-pool after each b-convolution.  The
// convolutions run in MIOpen WITHOUT bias; these kernels finish the layer in ONE pass over the
// activation instead of three (bias add, ReLU, pool):
//   bias_relu_kernel        x <- relu(x + bias[c])                                  (in place)
//   bias_pool_relu_kernel   y  = relu(max_pool2x2(x) + bias[c])   [B,C,H,W] -> [B,C,H/2,W/2]
// max(a,b)+c == max(a+c,b+c) and relu(max) == max(relu) hold exactly in floating point (monotone
// rounding), so the result is bit-identical to conv -> +bias -> ReLU -> max_pool.
// HBM-bound: the full-resolution activations are 3.2 GB per 16-pair batch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

__global__ void __launch_bounds__(256) bias_relu_kernel(float *__restrict__ x, const float *__restrict__ bias, int C, int HW)
{
    const int plane = blockIdx.y;                      // b * C + c
    const float bv = bias[plane % C];
    float *p = x + (size_t)plane * HW;
    const int n4 = HW >> 2;
    if (((size_t)p & 15) == 0) {
        float4 *p4 = (float4 *)p;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
            float4 v = p4[i];
            v.x = fmaxf(v.x + bv, 0.f); v.y = fmaxf(v.y + bv, 0.f); v.z = fmaxf(v.z + bv, 0.f); v.w = fmaxf(v.w + bv, 0.f);
            p4[i] = v;
        }
        for (int i = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) p[i] = fmaxf(p[i] + bv, 0.f);
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) p[i] = fmaxf(p[i] + bv, 0.f);
    }
}

// W % 4 == 0 fast path: each thread makes 2 outputs from two 16-byte loads
__global__ void __launch_bounds__(256) bias_pool_relu_v4_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                                int C, int H, int W, float *__restrict__ y)
{
    const int plane = blockIdx.y;
    const float bv = bias[plane % C];
    const int Ho = H >> 1, Wo = W >> 1, W4 = W >> 2;
    const float4 *p = (const float4 *)(x + (size_t)plane * H * W);
    float2 *o = (float2 *)(y + (size_t)plane * Ho * Wo);
    const int total = Ho * W4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int yo = i / W4, xq = i - yo * W4;
        const float4 a = p[(size_t)(2 * yo) * W4 + xq], b = p[(size_t)(2 * yo + 1) * W4 + xq];
        float2 r;
        r.x = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y)) + bv, 0.f);
        r.y = fmaxf(fmaxf(fmaxf(a.z, a.w), fmaxf(b.z, b.w)) + bv, 0.f);
        o[(size_t)yo * (Wo >> 1) + xq] = r;
    }
}

__global__ void __launch_bounds__(256) bias_pool_relu_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                             int C, int H, int W, float *__restrict__ y)
{
    const int plane = blockIdx.y;
    const float bv = bias[plane % C];
    const int Ho = H >> 1, Wo = W >> 1;
    const float *p = x + (size_t)plane * H * W;
    float *o = y + (size_t)plane * Ho * Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Ho * Wo; i += gridDim.x * 256) {
        const int yo = i / Wo, xo = i - yo * Wo;
        const float *r0 = p + (size_t)(2 * yo) * W + 2 * xo, *r1 = r0 + W;
        o[i] = fmaxf(fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r1[0], r1[1])) + bv, 0.f);
    }
}

extern "C" {

int mfr_bias_relu_nchw(float *x, const float *bias, int B, int C, int HW, void *stream)
{
    if (!x || !bias || B <= 0 || C <= 0 || HW <= 0) return MFR_E_ARG;
    int gx = (HW / 4 + 255) / 256; if (gx > 64) gx = 64; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(bias_relu_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, x, bias, C, HW);
    CHECK_LAUNCH();
    return 0;
}

int mfr_bias_pool2_relu_nchw(const float *x, const float *bias, int B, int C, int H, int W, float *y, void *stream)
{
    if (!x || !bias || !y || B <= 0 || C <= 0 || H < 2 || W < 2) return MFR_E_ARG;
    const int Ho = H / 2, Wo = W / 2;
    if ((W & 3) == 0 && (((size_t)x | (size_t)y) & 15) == 0) {
        int gx = (Ho * (W / 4) + 255) / 256; if (gx > 64) gx = 64;
        hipLaunchKernelGGL(bias_pool_relu_v4_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, x, bias, C, H, W, y);
    } else {
        int gx = (Ho * Wo + 255) / 256; if (gx > 64) gx = 64;
        hipLaunchKernelGGL(bias_pool_relu_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, x, bias, C, H, W, y);
    }
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
