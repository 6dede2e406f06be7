// loftr_fused.hip -- HBM-bound fusions of LoFTR's transformer / FPN glue (gfx950), f32.
//
// Reference call site: LoFTR_matcher.match (etc/feature_matching_baselines/matchers.py:24-59) -> upstream LoFTR
// (un-vendored zju3dv submodule; restated in SURVEY.md Appendix A.4):
//   LoFTREncoderLayer:   message = norm1(merge(attention));  message = mlp(cat[x, message]);  x + norm2(message)
//   ResNetFPN_8_2:       x2_out = layer2_outconv(x2) + interpolate(x3_out, scale 2, bilinear, align_corners=True)
// The library versions of these steps (LayerNorm 0.55 ms per call at 1/5 of the HBM rate, cat, interpolate + add as three
// full-size passes) were 16 % of the LoFTR step (profiles/r02a kernel stats).  Here:
//   layernorm_kernel      one wavefront per token row (C = 128 or 256: one float2 / float4 per lane), mean and variance by
//                         xor-butterfly, y = (x - mean) * rstd * gamma + beta [+ residual], input / residual / output with
//                         independent row strides -- so norm1 writes straight into the right half of the [x | message]
//                         operand of the MLP GEMM (no cat) and norm2 + residual updates x in place (no add kernel).
//                         Algorithmic bytes: C * 4 in + C * 4 out per row (+ C * 4 residual).
//   upsample2x_add_kernel y += bilinear_2x(lo), align_corners = True, NCHW: one pass over y instead of write(up) + read(up) +
//                         read(y) + write(y).  Same interpolation arithmetic as torch's upsample_bilinear2d (f32 lambdas).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int VEC>          // C = 64 * VEC
__global__ void __launch_bounds__(256) layernorm_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, const float *residual, int ldr, long long rows,
                                                        float eps, float *out, int ldo)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    typedef float vec __attribute__((ext_vector_type(VEC)));
    const vec v = *(const vec *)(x + row * ldx + lane * VEC);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.0f / (64 * VEC));
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / (64 * VEC)) + eps);
    const vec g = *(const vec *)(gamma + lane * VEC), b = *(const vec *)(beta + lane * VEC);
    vec y;
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = (v[i] - mean) * rstd * g[i] + b[i];
    if (residual) {
        const vec r = *(const vec *)(residual + row * ldr + lane * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) y[i] = r[i] + y[i];
    }
    *(vec *)(out + row * ldo + lane * VEC) = y;
}

// y [planes, 2H, 2W] += bilinear(lo [planes, H, W]), align_corners = True; each thread makes 4 consecutive outputs of a row
__global__ void __launch_bounds__(256) upsample2x_add_kernel(const float *__restrict__ lo, float *__restrict__ y, int H, int W, float rh, float rw)
{
    const int plane = blockIdx.z, oy = blockIdx.y;
    const int Wo = 2 * W, Ho = 2 * H;
    const int ox0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (ox0 >= Wo) return;
    const float h1r = rh * oy;
    const int h1 = (int)h1r;
    const int h1p = (h1 < H - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l;
    const float *r0 = lo + ((size_t)plane * H + h1) * W, *r1 = r0 + h1p * W;
    float *o = y + ((size_t)plane * Ho + oy) * Wo + ox0;
    float acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ox = ox0 + k;
        const float w1r = rw * ox;
        const int w1 = min((int)w1r, W - 1);
        const int w1p = (w1 < W - 1) ? 1 : 0;
        const float w1l = w1r - w1, w0l = 1.f - w1l;
        acc[k] = h0l * (w0l * r0[w1] + w1l * r0[w1 + w1p]) + h1l * (w0l * r1[w1] + w1l * r1[w1 + w1p]);
    }
    if (ox0 + 3 < Wo && (((size_t)o & 15) == 0)) {
        float4 v = *(float4 *)o;
        v.x += acc[0]; v.y += acc[1]; v.z += acc[2]; v.w += acc[3];
        *(float4 *)o = v;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (ox0 + k < Wo) o[k] += acc[k];
    }
}

extern "C" {

int mfr_layernorm(const float *x, int ldx, const float *gamma, const float *beta, const float *residual, int ldr, long long rows, int C,
                  float eps, float *out, int ldo, void *stream)
{
    if (!x || !gamma || !beta || !out || rows < 0 || (C != 128 && C != 256) || ldx < C || ldo < C || (residual && ldr < C)) return MFR_E_ARG;
    if ((ldx | ldo | (residual ? ldr : 0)) & 3) return MFR_E_ARG;                     // vector loads need 16-byte aligned rows
    if ((((size_t)x | (size_t)out | (size_t)gamma | (size_t)beta | (size_t)residual) & 15)) return MFR_E_ARG;
    if (rows == 0) return 0;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (C == 256) hipLaunchKernelGGL(layernorm_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, residual, ldr, rows, eps, out, ldo);
    else hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, residual, ldr, rows, eps, out, ldo);
    CHECK_LAUNCH();
    return 0;
}

int mfr_upsample2x_add(const float *lo, float *y, int planes, int H, int W, void *stream)
{
    if (!lo || !y || planes <= 0 || H <= 0 || W <= 0 || 2 * H > 65535 || planes > 65535 * 1) return MFR_E_ARG;
    const float rh = (2 * H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = (2 * W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const dim3 grid((2 * W + 1023) / 1024, 2 * H, planes);
    hipLaunchKernelGGL(upsample2x_add_kernel, grid, dim3(256), 0, (hipStream_t)stream, lo, y, H, W, rh, rw);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
