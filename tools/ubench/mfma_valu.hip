// Micro-benchmark: what does a VALU / SALU / DS instruction cost next to v_mfma_f32_16x16x4_f32 on gfx950?
// For each filler kind K and count N, a loop of {1 MFMA (4 independent accumulators round-robin) + N fillers}
// runs with 1 or 2 wavefronts per SIMD; reports cycles per MFMA group (s_memtime) -- 32 = fully hidden.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip ; run: ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int N>
__global__ void __launch_bounds__(256) probe(float *out, long long *cyc, int iters, const float *gsrc)
{
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float f[8];
    int g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = a + i; g[i] = threadIdx.x + i; }
    __shared__ __attribute__((aligned(16))) float lds[2048];
    lds[threadIdx.x] = a;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int r = (m * N + k) & 7;
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[r]) : "v"(b));
                if (KIND == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(g[r]) : "v"(g[(r + 1) & 7]));
                if (KIND == 2) asm volatile("s_nop 0");
                if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(g[r]) : "v"(g[(r + 1) & 7]));
                if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(g[r]) : "v"(g[(r + 1) & 7]));
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&f[(r & 3) * 2]) : "v"(*(double *)&f[((r + 1) & 3) * 2]));
                if (KIND == 6) { f32x4 q = *(volatile f32x4 *)&lds[(threadIdx.x & 63) * 4 + ((k & 1) << 8)]; f[r] += q[0]; }      // ds_read_b128 (+1 VALU)
                if (KIND == 7) { *(volatile f32x4 *)&lds[(threadIdx.x & 63) * 4 + ((k & 1) << 8)] = acc[k & 3]; }                 // ds_write_b128
                if (KIND == 8) { f[r] += __builtin_nontemporal_load(gsrc + ((it * 8 + m) * N + k) * 256 + threadIdx.x); }            // global_load_dword (+1 VALU)
                if (KIND == 9) { float q = *(volatile float *)&lds[(threadIdx.x + k) & 1023]; f[r] += q; }                          // ds_read_b32 (+1 VALU)
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + g[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int N>
void run(const char *name, int blocks_per_cu)
{
    const int iters = 500, nb = 256 * blocks_per_cu;
    float *out; long long *cyc;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static float *gsrc = nullptr; if (!gsrc) { hipMalloc(&gsrc, 64u << 20); hipMemset(gsrc, 0, 64u << 20); }
    probe<KIND, N><<<nb, 256>>>(out, cyc, 10, gsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, N><<<nb, 256>>>(out, cyc, iters, gsrc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // per SIMD: blocks_per_cu waves, each iters*8 MFMAs
    const double mfma_per_simd = (double)blocks_per_cu * iters * 8;
    printf("%-10s N=%d waves/SIMD=%d  wall %.3f ms  -> %.1f ns per MFMA-group per SIMD; s_memtime ticks per group (wave view) %.1f\n", name, N, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, (double)h[0] / (iters * 8));
    hipFree(out); hipFree(cyc);
}

#define ROW(K, NAME) run<K, 0>(NAME, w); run<K, 1>(NAME, w); run<K, 2>(NAME, w); run<K, 3>(NAME, w); run<K, 4>(NAME, w); run<K, 6>(NAME, w); run<K, 8>(NAME, w);
int main()
{
    for (int w = 1; w <= 2; ++w) {
        ROW(0, "v_add_f32") ROW(6, "ds_rd128+1") ROW(7, "ds_wr128") ROW(8, "gload+1") ROW(9, "ds_rd32+1")
    }
    return 0;
}
