// bf16x3_probe.hip -- gate for the "fp32-accurate matrix products on the bf16 matrix cores" path:
//   (1) pins the A / B / C lane layouts of v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16 against a host product;
//   (2) measures the error of a 3-way bf16 operand split (x = h + m + l, 6 products hh, hm, mh, hl, lh, mm, f32 accumulate)
//       against an fp64 product, next to the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) on the same data;
//   (3) measures how many VALU instructions fit beside a back-to-back bf16 MFMA stream (1 and 2 waves per SIMD).
// build: hipcc -O3 --offload-arch=gfx950 -o bf16x3_probe bf16x3_probe.hip ; run on the GPU box, prints JSON lines.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned short bf16_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ float bf16_up(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// 3-way split by truncation: exact, h + m + l == x for every finite fp32 x (8 + 8 + 8 significand bits)
__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l)
{
    h = bf16_trunc(x); const float r = x - bf16_up(h);
    m = bf16_trunc(r); const float s = r - bf16_up(m);
    l = bf16_trunc(s);
}

// ---------------------------------------------------------------- (1) layouts
__global__ void layout32(const float *A /*[32][16]*/, const float *B /*[16][32]*/, float *C /*[32][32]*/)
{
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)bf16_trunc(A[(l & 31) * 16 + 8 * (l >> 5) + j]);
        b[j] = (short)bf16_trunc(B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void layout16(const float *A /*[16][32]*/, const float *B /*[32][16]*/, float *C /*[16][16]*/)
{
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)bf16_trunc(A[(l & 15) * 32 + 8 * (l >> 4) + j]);
        b[j] = (short)bf16_trunc(B[(8 * (l >> 4) + j) * 16 + (l & 15)]);
    }
    f32x4 c = { 0.f, 0.f, 0.f, 0.f };
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// ---------------------------------------------------------------- (2) accuracy: C[32][32] = A[32][K] B[K][32], one wave per problem
// mode 0: exact fp32 MFMA; 1: 6 products, small terms first; 2: 6 products, large first; 3: 3 products (hh, hm, mh); 4: 9 products
__global__ void acc_kernel(const float *A, const float *B, float *C, int K, int mode)
{
    const int l = threadIdx.x, p = blockIdx.x;
    A += (size_t)p * 32 * K; B += (size_t)p * K * 32; C += (size_t)p * 1024;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k + (l >> 5)], B[(k + (l >> 5)) * 32 + (l & 31)], c, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                unsigned short h, m, lo;
                split3(A[(l & 31) * K + k + 8 * (l >> 5) + j], h, m, lo); a[0][j] = (short)h; a[1][j] = (short)m; a[2][j] = (short)lo;
                split3(B[(k + 8 * (l >> 5) + j) * 32 + (l & 31)], h, m, lo); b[0][j] = (short)h; b[1][j] = (short)m; b[2][j] = (short)lo;
            }
#define MM(i, j) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], c, 0, 0, 0)
            if (mode == 1) { MM(1, 1); MM(0, 2); MM(2, 0); MM(0, 1); MM(1, 0); MM(0, 0); }
            else if (mode == 2) { MM(0, 0); MM(0, 1); MM(1, 0); MM(0, 2); MM(2, 0); MM(1, 1); }
            else if (mode == 3) { MM(0, 1); MM(1, 0); MM(0, 0); }
            else { MM(2, 2); MM(1, 2); MM(2, 1); MM(1, 1); MM(0, 2); MM(2, 0); MM(0, 1); MM(1, 0); MM(0, 0); }
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// ---------------------------------------------------------------- (3) VALU beside a bf16 MFMA stream
template <int NV>
__global__ void __launch_bounds__(256) overlap_kernel(float *out, int iters, float seed)
{
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + threadIdx.x * 0.001f + j;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {                          // the split's instruction mix: and / sub / perm-like ops
            const float h = __uint_as_float(__float_as_uint(v[j & 7]) & 0xffff0000u);
            v[j & 7] = (v[j & 7] - h) * 1.0001f + h;
        }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
static void run_overlap(int waves_per_simd)
{
    const int iters = 4000, blocks = 256 * waves_per_simd;      // 256-thread blocks = 4 waves = one per SIMD
    float *out; CK(hipMalloc(&out, sizeof(float) * 256 * blocks));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(overlap_kernel<NV>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(overlap_kernel<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma = 4.0 * iters * 4 * blocks;               // per SIMD stream: 4 per iteration per wave
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    // NV "units" = 3 VALU each (and, sub, fma-ish mul+add folded by the compiler: counted from the source as ~4)
    printf("{\"probe\": \"overlap\", \"waves_per_simd\": %d, \"valu_units_per_4_mfma\": %d, \"ms\": %.3f, \"bf16_tflops\": %.1f, \"ns_per_mfma_per_simd\": %.2f}\n",
           waves_per_simd, NV, ms, tf, ms * 1e6 / (4.0 * iters * waves_per_simd));
    CK(hipFree(out));
}

int main()
{
    // (1)
    {
        std::vector<float> A(512), B(512), C(1024), R(1024);
        for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + 3) % 11 - 5); B[i] = (float)((i * 5 + 1) % 13 - 6); }
        float *dA, *dB, *dC; CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 4096));
        CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; bad += (s != C[i * 32 + j]); }
        printf("{\"probe\": \"layout_32x32x16_bf16\", \"mismatches\": %d}\n", bad);
        hipLaunchKernelGGL(layout16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
        bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; bad += (s != C[i * 16 + j]); }
        printf("{\"probe\": \"layout_16x16x32_bf16\", \"mismatches\": %d}\n", bad);
    }
    // (2)
    for (int K : { 64, 256, 512, 2304 }) {
        const int P = 64;
        std::vector<float> A((size_t)P * 32 * K), B((size_t)P * K * 32), C((size_t)P * 1024);
        srand(1234 + K);
        auto rnd = [] { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };
        for (auto &x : A) x = rnd(); for (auto &x : B) x = rnd() * 0.05f;
        float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> R((size_t)P * 1024), S((size_t)P * 1024);
        for (int p = 0; p < P; ++p) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { double t = (double)A[((size_t)p * 32 + i) * K + k] * (double)B[((size_t)p * K + k) * 32 + j]; s += t; sa += fabs(t); }
            R[(size_t)p * 1024 + i * 32 + j] = s; S[(size_t)p * 1024 + i * 32 + j] = sa;
        }
        const char *names[5] = { "fp32_mfma_exact", "bf16x3_6prod_small_first", "bf16x3_6prod_large_first", "bf16x3_3prod", "bf16x3_9prod" };
        for (int mode = 0; mode < 5; ++mode) {
            hipLaunchKernelGGL(acc_kernel, dim3(P), dim3(64), 0, 0, dA, dB, dC, K, mode);
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double mx = 0, sq = 0, mxr = 0;
            for (size_t i = 0; i < C.size(); ++i) { double e = fabs((double)C[i] - R[i]) / S[i]; mx = fmax(mx, e); sq += e * e; mxr = fmax(mxr, fabs((double)C[i] - R[i])); }
            printf("{\"probe\": \"accuracy\", \"K\": %d, \"mode\": \"%s\", \"max_err_over_sum_abs\": %.3e, \"rms_err_over_sum_abs\": %.3e, \"max_abs_err\": %.3e}\n",
                   K, names[mode], mx, sqrt(sq / C.size()), mxr);
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    // (3)
    for (int w = 1; w <= 2; ++w) { run_overlap<0>(w); run_overlap<2>(w); run_overlap<4>(w); run_overlap<8>(w); run_overlap<16>(w); run_overlap<24>(w); run_overlap<32>(w); }
    return 0;
}
