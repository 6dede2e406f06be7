// f16x2_probe.hip -- gate for the "f16x2" operand split (map-free-reloc_amd/csrc/split_f16.h) before any kernel relies on it:
//   (1) lane layouts of v_mfma_f32_32x32x16_f16 against a host product;
//   (2) subnormal f16 A / B inputs of that instruction are NOT flushed (the scheme's tiny terms are subnormal f16 numbers);
//   (3) the device split (v_cvt_pk_f16_f32 + v_fma_mix_f32 + v_fma_mixlo/hi_f16) equals the host definition bit for bit and
//       reproduces x to 2^-24 |x| on 2^-14 <= |x| <= 65504;
//   (4) error of the 3-product f16x2 scheme against an fp64 product, next to the exact-fp32 MFMA and the 6-product bf16x3 split on the SAME
//       data, at activation scales 1, 1e-3, 1e3 (the bf16x3 / fp32 paths are scale-free; f16x2 has a finite exponent range);
//   (5) VALU instructions beside a back-to-back f16 MFMA stream (the split's instruction mix).
// build: hipcc -O3 --offload-arch=gfx950 -o f16x2_probe f16x2_probe.hip ; run on the GPU box, prints JSON lines.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../map-free-reloc_amd/csrc/split_f16.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static float h2f(unsigned short h)
{
    const unsigned s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 1024), (int)e - 25);
    return s ? -v : v;
}
static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }   // host RNE conversion

// ---------------------------------------------------------------- (1) layout
__global__ void layout32(const float *A /*[32][16]*/, const float *B /*[16][32]*/, float *C /*[32][32]*/)
{
    const int l = threadIdx.x;
    sf_f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (_Float16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// ---------------------------------------------------------------- (2) subnormal inputs: A = 2^-20 (subnormal f16), B = 1 -> C = 16 * 2^-20; A = 2^-20, B = 2^-20 -> 16 * 2^-40
__global__ void subnormal(float *out)
{
    sf_f16x8 a, b, one;
    const _Float16 tiny = __builtin_bit_cast(_Float16, (unsigned short)0x0010);      // 16 * 2^-24 = 2^-20
    for (int j = 0; j < 8; ++j) { a[j] = tiny; b[j] = tiny; one[j] = (_Float16)1.0f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    f32x16 c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, one, c, 0, 0, 0);
    f32x16 c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(one, b, c, 0, 0, 0);
    f32x16 c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c1[0]; out[1] = c2[0]; out[2] = c3[0]; }
}

// ---------------------------------------------------------------- (3) the split
__global__ void split_kernel(const float *x, unsigned *h, unsigned *l, int n2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    unsigned hh, ll;
    sf_split2(x[2 * i], x[2 * i + 1], SF_LOW_SCALE, hh, ll);
    h[i] = hh; l[i] = ll;
}

// ---------------------------------------------------------------- (4) accuracy: C[32][32] = A[32][K] B[K][32], one wave per problem
// mode 0: exact fp32 MFMA; 1: bf16x3, 6 products; 2: f16x2 (A = activation side, B = weight side with per-column scale)
__device__ __forceinline__ unsigned short bf16_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ float bf16_up(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l)
{
    h = bf16_trunc(x); const float r = x - bf16_up(h);
    m = bf16_trunc(r); const float s = r - bf16_up(m);
    l = bf16_trunc(s);
}
__global__ void acc_kernel(const float *A, const float *B, float *C, int K, int mode)
{
    const int l = threadIdx.x, p = blockIdx.x;
    A += (size_t)p * 32 * K; B += (size_t)p * K * 32; C += (size_t)p * 1024;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    float inv = 1.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k + (l >> 5)], B[(k + (l >> 5)) * 32 + (l & 31)], c, 0, 0, 0);
    } else if (mode == 1) {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                unsigned short h, m, lo;
                split3(A[(l & 31) * K + k + 8 * (l >> 5) + j], h, m, lo); a[0][j] = (short)h; a[1][j] = (short)m; a[2][j] = (short)lo;
                split3(B[(k + 8 * (l >> 5) + j) * 32 + (l & 31)], h, m, lo); b[0][j] = (short)h; b[1][j] = (short)m; b[2][j] = (short)lo;
            }
#define MM(i, j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], c, 0, 0, 0)
            MM(1, 1); MM(0, 2); MM(2, 0); MM(0, 1); MM(1, 0); MM(0, 0);
#undef MM
        }
    } else {
        // weight column (l & 31): scale from its maximum
        float mx = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(B[k * 32 + (l & 31)]));
        const float s = sf_feature_scale(mx);
        for (int k = 0; k < K; k += 16) {
            unsigned ah[4], al[4];
            unsigned short bh[8], bl[8], bq[8];
            for (int j = 0; j < 4; ++j) sf_split2(A[(l & 31) * K + k + 8 * (l >> 5) + 2 * j], A[(l & 31) * K + k + 8 * (l >> 5) + 2 * j + 1], SF_LOW_SCALE, ah[j], al[j]);
            for (int j = 0; j < 8; ++j) sf_split_w(B[(k + 8 * (l >> 5) + j) * 32 + (l & 31)] * s, bh[j], bl[j], bq[j]);
            uint4 xh = make_uint4(ah[0], ah[1], ah[2], ah[3]), xl = make_uint4(al[0], al[1], al[2], al[3]);
            uint4 wh = make_uint4(bh[0] | (unsigned)bh[1] << 16, bh[2] | (unsigned)bh[3] << 16, bh[4] | (unsigned)bh[5] << 16, bh[6] | (unsigned)bh[7] << 16);
            uint4 wl = make_uint4(bl[0] | (unsigned)bl[1] << 16, bl[2] | (unsigned)bl[3] << 16, bl[4] | (unsigned)bl[5] << 16, bl[6] | (unsigned)bl[7] << 16);
            uint4 wq = make_uint4(bq[0] | (unsigned)bq[1] << 16, bq[2] | (unsigned)bq[3] << 16, bq[4] | (unsigned)bq[5] << 16, bq[6] | (unsigned)bq[7] << 16);
            c = SF_MFMA(xl, wq, c); c = SF_MFMA(xh, wl, c); c = SF_MFMA(xh, wh, c);
        }
        inv = 1.f / s;
    }
    // accumulator (r, lane): row (A index) (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (B index) lane & 31 -> the column's scale is the lane's
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r] * inv;
}

// ---------------------------------------------------------------- (5) the split beside an f16 MFMA stream
template <int NV>
__global__ void __launch_bounds__(256) overlap_kernel(float *out, int iters, float seed)
{
    sf_f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(1.0f + 0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.5f + 0.01f * j); }
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + threadIdx.x * 0.001f + j;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {                          // one unit = the split of two elements: 5 VALU
            unsigned h, l;
            sf_split2(v[(2 * j) & 7], v[(2 * j + 1) & 7], SF_LOW_SCALE, h, l);
            acc ^= h ^ l;
            v[(2 * j) & 7] += 1.0f;
        }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    float s = (float)acc;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
static void run_overlap(int waves_per_simd)
{
    const int iters = 4000, blocks = 256 * waves_per_simd;
    float *out; CK(hipMalloc(&out, sizeof(float) * 256 * blocks));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(overlap_kernel<NV>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(overlap_kernel<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = 4.0 * iters * 4 * blocks;
    printf("{\"probe\": \"overlap_f16\", \"waves_per_simd\": %d, \"split_pairs_per_4_mfma\": %d, \"valu_per_mfma\": %.2f, \"ms\": %.3f, \"f16_tflops\": %.1f, \"ns_per_mfma_per_simd\": %.2f}\n",
           waves_per_simd, NV, (7.0 * NV) / 4.0, ms, mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12, ms * 1e6 / (4.0 * iters * waves_per_simd));
    CK(hipFree(out));
}

int main()
{
    // (1)
    {
        std::vector<float> A(512), B(512), C(1024);
        for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + 3) % 11 - 5); B[i] = (float)((i * 5 + 1) % 13 - 6); }
        float *dA, *dB, *dC; CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; bad += (s != C[i * 32 + j]); }
        printf("{\"probe\": \"layout_32x32x16_f16\", \"mismatches\": %d}\n", bad);
    }
    // (2)
    {
        float *d; CK(hipMalloc(&d, 16)); float h[3];
        hipLaunchKernelGGL(subnormal, dim3(1), dim3(64), 0, 0, d);
        CK(hipMemcpy(h, d, 12, hipMemcpyDeviceToHost));
        printf("{\"probe\": \"subnormal_f16_inputs\", \"a_sub_x_one\": %.6e, \"one_x_b_sub\": %.6e, \"expected\": %.6e, \"sub_x_sub\": %.6e, \"expected_sub_x_sub\": %.6e, \"kept\": %s}\n",
               h[0], h[1], 16.0 * ldexp(1.0, -20), h[2], 16.0 * ldexp(1.0, -40),
               (h[0] == 16.0f * ldexpf(1.f, -20) && h[1] == h[0] && h[2] == 16.0f * ldexpf(1.f, -40)) ? "true" : "false");
    }
    // (3)
    {
        const int n = 1 << 20;
        std::vector<float> x(n);
        srand(7);
        for (int i = 0; i < n; ++i) {
            const int e = (rand() % 44) - 26;                                   // magnitudes 2^-26 .. 2^17
            float m = 1.0f + rand() / (float)RAND_MAX; if (rand() & 1) m = -m;
            x[i] = ldexpf(m, e);
            if (fabsf(x[i]) > 65000.f) x[i] = 65000.f;
        }
        x[0] = 0.f; x[1] = -0.f; x[2] = 65504.f; x[3] = 6.1035156e-05f; x[4] = 5.9604645e-08f; x[5] = 1.0f; x[6] = 0.3333333f; x[7] = -1e-7f;
        float *dx; unsigned *dh, *dl; CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dh, n * 2)); CK(hipMalloc(&dl, n * 2));
        CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dh, dl, n / 2);
        std::vector<unsigned short> h(n), l(n);
        CK(hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(l.data(), dl, n * 2, hipMemcpyDeviceToHost));
        long bad_bits = 0; double worst_rel = 0, worst_abs_small = 0; long nsub = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned short eh = f2h(x[i]);
            const float r = x[i] - h2f(eh);
            const unsigned short el = f2h(r * 2048.0f);
            bad_bits += (eh != h[i]) || (el != l[i]);
            const double rep = (double)h2f(h[i]) + (double)h2f(l[i]) / 2048.0;
            const double err = fabs(rep - (double)x[i]);
            if (fabs(x[i]) >= 6.1035156e-05) worst_rel = fmax(worst_rel, err / fabs((double)x[i]));
            else worst_abs_small = fmax(worst_abs_small, err);
            nsub += ((l[i] & 0x7c00) == 0 && (l[i] & 0x3ff) != 0);
        }
        printf("{\"probe\": \"split\", \"n\": %d, \"bit_mismatches_vs_host_definition\": %ld, \"worst_rel_err_normal_range\": %.3e, \"two_pow_minus_24\": %.3e, \"worst_abs_err_below_2^-14\": %.3e, \"subnormal_low_terms\": %ld}\n",
               n, bad_bits, worst_rel, ldexp(1.0, -24), worst_abs_small, nsub);
    }
    // (4)
    for (float scale : { 1.0f, 1e-3f, 1e3f })
    for (int K : { 64, 512, 2304 }) {
        const int P = 64;
        std::vector<float> A((size_t)P * 32 * K), B((size_t)P * K * 32), C((size_t)P * 1024);
        srand(1234 + K);
        auto rnd = [] { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };
        for (auto &x : A) x = rnd() * scale; for (auto &x : B) x = rnd() * 0.05f;
        float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> R((size_t)P * 1024), S((size_t)P * 1024);
        for (int p = 0; p < P; ++p) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { double t = (double)A[((size_t)p * 32 + i) * K + k] * (double)B[((size_t)p * K + k) * 32 + j]; s += t; sa += fabs(t); }
            R[(size_t)p * 1024 + i * 32 + j] = s; S[(size_t)p * 1024 + i * 32 + j] = sa;
        }
        const char *names[3] = { "fp32_mfma_exact", "bf16x3_6prod", "f16x2_3prod" };
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(acc_kernel, dim3(P), dim3(64), 0, 0, dA, dB, dC, K, mode);
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double mx = 0, sq = 0;
            for (size_t i = 0; i < C.size(); ++i) { double e = fabs((double)C[i] - R[i]) / S[i]; mx = fmax(mx, e); sq += e * e; }
            printf("{\"probe\": \"accuracy\", \"activation_scale\": %g, \"K\": %d, \"mode\": \"%s\", \"max_err_over_sum_abs\": %.3e, \"rms_err_over_sum_abs\": %.3e}\n",
                   scale, K, names[mode], mx, sqrt(sq / C.size()));
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    // (5)
    for (int w = 1; w <= 2; ++w) { run_overlap<0>(w); run_overlap<1>(w); run_overlap<2>(w); run_overlap<3>(w); run_overlap<4>(w); run_overlap<6>(w); run_overlap<8>(w); }
    return 0;
}
