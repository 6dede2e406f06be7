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

// First SuperPoint layer fused: y = relu(conv3x3(x, w) + b) for ONE input channel and 64 output
// channels (conv1a: 0.45 GFLOP per image against 100 MB of output -> purely HBM-write-bound).  MIOpen
// Winograd + a separate bias/ReLU pass took 2.7 ms per 32 images; one pass writing 3.2 GB is ~0.6 ms.
// Each thread owns 4 adjacent pixels (3x6 input patch in registers); weights are wave-uniform.
#define C1_OC 64
__global__ void __launch_bounds__(256) conv3x3_c1_relu_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, int H, int W,
                                                              float *__restrict__ y)
{
    __shared__ float sw[C1_OC * 9 + C1_OC];
    for (int i = threadIdx.x; i < C1_OC * 9; i += 256) sw[i] = w[i];
    for (int i = threadIdx.x; i < C1_OC; i += 256) sw[C1_OC * 9 + i] = bias[i];
    __syncthreads();
    const int b = blockIdx.y, W4 = W >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W4) return;
    const int py = t / W4, px = (t - py * W4) * 4;
    const float *img = x + (size_t)b * H * W;
    float p[3][6];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = py + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 6; ++dx) {
            const int xx = px + dx - 1;
            p[dy][dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(size_t)yy * W + xx] : 0.f;
        }
    }
    float *o = y + (size_t)b * C1_OC * H * W + (size_t)py * W + px;
    for (int c = 0; c < C1_OC; ++c) {
        const float *k = sw + 9 * c;
        const float bv = sw[C1_OC * 9 + c];
        float4 r;
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = fmaf(k[3 * dy + dx], p[dy][j + dx], a);
            acc[j] = fmaxf(a + bv, 0.f);
        }
        r.x = acc[0]; r.y = acc[1]; r.z = acc[2]; r.w = acc[3];
        *(float4 *)(o + (size_t)c * H * W) = r;
    }
}


// 1x1 convolution with few output channels, NCHW f32 (SuperPoint's detector head convPb, 256 -> 65): y[b, co, p] = bias[co] +
// sum_c w[co, c] x[b, c, p], one thread per pixel, 16 output channels per workgroup row, the weights of the block broadcast from LDS,
// ONE fused-multiply-add chain over c in ascending order per output -- so a pixel's logits do not depend on how many images are in the
// batch (the library's batched GEMM picked its tile, and with it the summation order, from the batch size: the reference-view feature
// cache of pipeline.py needs per-image results that are the same bits in a batch of 33 and in a batch of 64).
#define C1_CB 16
__global__ void __launch_bounds__(256) conv1x1_nchw_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                           int Cin, int Cout, int HW, float *__restrict__ y)
{
    __shared__ float ws[512 * C1_CB];                       // [c][k]
    const int b = blockIdx.z, cb0 = blockIdx.y * C1_CB, p = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < Cin * C1_CB; i += 256) {
        const int c = i / C1_CB, k = i - c * C1_CB;
        ws[i] = (cb0 + k < Cout) ? w[(size_t)(cb0 + k) * Cin + c] : 0.f;
    }
    __syncthreads();
    if (p >= HW) return;
    float acc[C1_CB];
#pragma unroll
    for (int k = 0; k < C1_CB; ++k) acc[k] = (bias && cb0 + k < Cout) ? bias[cb0 + k] : 0.f;
    const float *xp = x + (size_t)b * Cin * HW + p;
    for (int c = 0; c < Cin; ++c) {
        const float xv = xp[(size_t)c * HW];
        const float4 *wr = (const float4 *)(ws + c * C1_CB);
#pragma unroll
        for (int q = 0; q < C1_CB / 4; ++q) {
            const float4 wv = wr[q];
            acc[4 * q] = __builtin_fmaf(wv.x, xv, acc[4 * q]); acc[4 * q + 1] = __builtin_fmaf(wv.y, xv, acc[4 * q + 1]);
            acc[4 * q + 2] = __builtin_fmaf(wv.z, xv, acc[4 * q + 2]); acc[4 * q + 3] = __builtin_fmaf(wv.w, xv, acc[4 * q + 3]);
        }
    }
    float *yp = y + ((size_t)b * Cout + cb0) * HW + p;
#pragma unroll
    for (int k = 0; k < C1_CB; ++k)
        if (cb0 + k < Cout) yp[(size_t)k * HW] = acc[k];
}

// NCHW feature maps -> token-major rows: out[img'][p][c] = x[img][c][p] (+ add[c][p]); 64 x 64 tiles through LDS (row stride 65: both the
// pixel-contiguous reads and the channel-contiguous writes are conflict-free and 256-byte coalesced).  Replaces torch's strided copy
// (permute().contiguous(): 0.50 ms for SuperPoint's 64 x 256 x 90 x 67 descriptor map, 1.6 TB/s) in front of the linear layers that
// consume rows.  deinterleave: image 2 k + s of a pair-interleaved batch lands at s * (B / 2) + k (LoFTR's side-major token buffer).
__global__ void __launch_bounds__(256) nchw_to_rows_kernel(const float *__restrict__ x, const float *__restrict__ add, int C, int HW, int B, int deinterleave,
                                                           float *__restrict__ out, long long out_img_stride, int ldo)
{
    __shared__ float t[64][65];
    const int img = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float *xi = x + (size_t)img * C * HW;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + ty + 4 * k, pp = p0 + tx;
        float v = 0.f;
        if (c < C && pp < HW) { v = xi[(size_t)c * HW + pp]; if (add) v += add[(size_t)c * HW + pp]; }
        t[ty + 4 * k][tx] = v;
    }
    __syncthreads();
    const int oimg = deinterleave ? (img & 1) * (B >> 1) + (img >> 1) : img;
    float *oi = out + (size_t)oimg * out_img_stride;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int pp = p0 + ty + 4 * k, c = c0 + tx;
        if (pp < HW && c < C) oi[(size_t)pp * ldo + c] = t[tx][ty + 4 * k];
    }
}

#include "guard.h"
// the f16x2 range guard's flag pointer (guard.h): per host thread, NULL = no test
static thread_local int *mfr_guard_tls = nullptr;

extern "C" {

int *mfr_guard_current(void) { return mfr_guard_tls; }

int mfr_f16x2_guard_bind(int *device_flag)
{
    mfr_guard_tls = device_flag;
    return 0;
}


int mfr_conv3x3_c1_relu(const float *x, const float *w, const float *bias, int B, int H, int W, int out_channels,
                        float *y, void *stream)
{
    if (!x || !w || !bias || !y || B <= 0 || H <= 0 || W <= 0 || out_channels != C1_OC || (W & 3) ||
        (((size_t)y) & 15)) return MFR_E_ARG;
    hipLaunchKernelGGL(conv3x3_c1_relu_kernel, dim3((H * (W / 4) + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       H, W, y);
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv1x1_nchw(const float *x, const float *w, const float *bias, int B, int Cin, int Cout, int HW, float *y, void *stream)
{
    if (!x || !w || !y || B <= 0 || Cin <= 0 || Cin > 512 || Cout <= 0 || HW <= 0 || B > 65535) return MFR_E_ARG;
    hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3((HW + 255) / 256, (Cout + C1_CB - 1) / C1_CB, B), dim3(256), 0, (hipStream_t)stream, x, w, bias, Cin, Cout, HW, y);
    CHECK_LAUNCH();
    return 0;
}

int mfr_nchw_to_rows(const float *x, const float *add, int B, int C, int HW, int deinterleave, float *out, long long out_img_stride, int ldo, void *stream)
{
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0 || ldo < C || B > 65535 || (deinterleave && (B & 1))) return MFR_E_ARG;
    hipLaunchKernelGGL(nchw_to_rows_kernel, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, add, C, HW, B, deinterleave, out,
                       out_img_stride, ldo);
    CHECK_LAUNCH();
    return 0;
}

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
