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

// Bilinear resampling with align_corners = True, torch's arithmetic (upsample_bilinear2d: ratio = (in - 1) / (out - 1) in f32,
// source = ratio * dst, the four taps weighted in f32).  One thread makes 4 consecutive outputs of a row; the (plane, row, quad)
// index space is FLAT, so small maps (a 34-wide row is 17 quads) still fill whole wavefronts -- PyTorch's own kernel assigns
// one thread per output PIXEL and loops over batch x channels (6256 threads for the regression decoder's 92x68 map: 2.4 ms per
// call where the data moves in 40 us).
//   ADD:  y += bilinear(lo)   (LoFTR's FPN merge, in place, f32)
//   !ADD: y  = bilinear(lo)   (the regression decoder's upconv; f32 or bf16 storage, f32 arithmetic, round-to-nearest-even)
template <typename T> struct UpIO;
template <> struct UpIO<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct UpIO<unsigned short> {                                   // bfloat16 storage
    static __device__ __forceinline__ float ld(const unsigned short *p) { return __uint_as_float((unsigned)(*p) << 16); }
    static __device__ __forceinline__ void st(unsigned short *p, float v)
    {
        unsigned u = __float_as_uint(v);
        if ((u & 0x7fffffffu) > 0x7f800000u) { *p = (unsigned short)((u >> 16) | 0x40); return; }   // NaN stays NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        *p = (unsigned short)(u >> 16);
    }
};

template <typename T, bool ADD>
__global__ void __launch_bounds__(256) upsample_ac_kernel(const T *__restrict__ lo, T *__restrict__ y, int H, int W, int Ho, int Wo,
                                                          float rh, float rw, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int Q = (Wo + 3) >> 2;
    const int q = (int)(idx % Q);
    const size_t t = idx / Q;
    const int oy = (int)(t % Ho);
    const size_t plane = t / Ho;
    const int ox0 = q * 4;
    const float h1r = rh * oy;
    const int h1 = min((int)h1r, H - 1);
    const int h1p = (h1 < H - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l;
    const T *r0 = lo + (plane * H + h1) * W, *r1 = r0 + h1p * W;
    T *o = y + (plane * Ho + oy) * Wo + ox0;
    float acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ox = min(ox0 + k, Wo - 1);
        const float w1r = rw * ox;
        const int w1 = min((int)w1r, W - 1);
        const int w1p = (w1 < W - 1) ? 1 : 0;
        const float w1l = w1r - w1, w0l = 1.f - w1l;
        acc[k] = h0l * (w0l * UpIO<T>::ld(r0 + w1) + w1l * UpIO<T>::ld(r0 + w1 + w1p)) +
                 h1l * (w0l * UpIO<T>::ld(r1 + w1) + w1l * UpIO<T>::ld(r1 + w1 + w1p));
    }
    if constexpr (sizeof(T) == 4) {
        if (ox0 + 3 < Wo && (((size_t)o & 15) == 0)) {
            float4 v = ADD ? *(float4 *)o : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x += acc[0]; v.y += acc[1]; v.z += acc[2]; v.w += acc[3];
            *(float4 *)o = v;
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (ox0 + k < Wo) UpIO<T>::st(o + k, ADD ? UpIO<T>::ld(o + k) + acc[k] : acc[k]);
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

static inline float up_ratio(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

int mfr_upsample2x_add(const float *lo, float *y, int planes, int H, int W, void *stream)
{
    if (!lo || !y || planes <= 0 || H <= 0 || W <= 0) return MFR_E_ARG;
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)planes * Ho * ((Wo + 3) / 4);
    if ((total + 255) / 256 > 0x7fffffffull) return MFR_E_ARG;
    hipLaunchKernelGGL((upsample_ac_kernel<float, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, lo, y, H, W,
                       Ho, Wo, up_ratio(H, Ho), up_ratio(W, Wo), total);
    CHECK_LAUNCH();
    return 0;
}

// out [planes, Ho, Wo] = bilinear(in [planes, H, W]), align_corners = True; dtype 0: float32, 1: bfloat16 (both tensors)
int mfr_upsample_bilinear(const void *in, void *out, int planes, int H, int W, int Ho, int Wo, int dtype, void *stream)
{
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || (dtype != 0 && dtype != 1)) return MFR_E_ARG;
    const size_t total = (size_t)planes * Ho * ((Wo + 3) / 4);
    if ((total + 255) / 256 > 0x7fffffffull) return MFR_E_ARG;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == 0)
        hipLaunchKernelGGL((upsample_ac_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)out, H, W, Ho, Wo,
                           up_ratio(H, Ho), up_ratio(W, Wo), total);
    else
        hipLaunchKernelGGL((upsample_ac_kernel<unsigned short, false>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)in,
                           (unsigned short *)out, H, W, Ho, Wo, up_ratio(H, Ho), up_ratio(W, Wo), total);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
