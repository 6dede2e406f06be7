// Micro-benchmark (round 4): what does a VALU instruction cost next to v_mfma_f32_32x32x16_bf16 on gfx950, from the SAME wavefront and
// from the OTHER wavefront of the SIMD?  Loop of {1 MFMA (4 independent accumulators round-robin) + N VALU fillers (the and / sub / perm
// mix of the bf16x3 operand split)}, 1 or 2 wavefronts per SIMD; mode "split": with 2 wavefronts per SIMD one issues only the MFMAs and
// the other only the fillers of both.  Reports ns per MFMA group per SIMD (32 cycles at the run's clock = fully hidden).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_bf16 mfma_valu_bf16.hip ; run: ./mfma_valu_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int MODE>   // MODE 0: every wavefront does MFMA + N fillers; 1: even blocks MFMA only, odd blocks 2 N fillers per (absent) MFMA
__global__ void __launch_bounds__(256, 2) probe(float *out, int iters)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    unsigned g[8]; float f[8];
    for (int i = 0; i < 8; ++i) { g[i] = threadIdx.x * 2654435761u + i; f[i] = 1.0f + threadIdx.x * 1e-3f + i; }
    const bool do_mfma = (MODE == 0) || !(blockIdx.x & 256), do_valu = (MODE == 0) || (blockIdx.x & 256);
    const int nf = (MODE == 0) ? N : 2 * N;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (do_mfma) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int k = 0; k < (MODE == 0 ? N : 2 * N); ++k) {
                    const int r = (m * nf + k) & 7;
                    if ((k % 3) == 0) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(g[r]) : "v"(f[(r + 1) & 7]));
                    else if ((k % 3) == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[r]) : "v"(f[(r + 3) & 7]), "v"(g[(r + 5) & 7]));
                    else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(g[r]) : "v"(g[(r + 2) & 7]), "v"(f[(r + 4) & 7]), "s"(0x07060302u));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)g[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int MODE>
void run(int blocks_per_cu)
{
    const int iters = 2000, nb = 256 * blocks_per_cu;
    float *out;
    hipMalloc(&out, nb * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<N, MODE><<<nb, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<N, MODE><<<nb, 256>>>(out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // MFMA groups per SIMD: MODE 0: blocks_per_cu wavefronts x iters x 8; MODE 1: one MFMA wavefront per SIMD x iters x 8 (the other issues 2N fillers per group)
    const double groups = (MODE == 0 ? blocks_per_cu : 1) * (double)iters * 8;
    printf("{\"mode\": \"%s\", \"valu_per_mfma\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_mfma_group_per_simd\": %.2f, \"bf16_tflops\": %.0f}\n",
           MODE ? "split" : "same", MODE ? 2 * N : N, blocks_per_cu, ms, ms * 1e6 / groups, 1024.0 * groups * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
#define ROW(M, W) run<0, M>(W); run<2, M>(W); run<4, M>(W); run<5, M>(W); run<6, M>(W); run<8, M>(W); run<12, M>(W);
    ROW(0, 1) ROW(0, 2)
    run<1, 1>(2); run<2, 1>(2); run<3, 1>(2); run<4, 1>(2); run<6, 1>(2);
    return 0;
}
