// Micro-benchmark (round 4): the ceiling of an LDS-fed v_mfma_f32_32x32x16_bf16 loop on gfx950, built up part by part.  One "K step" of
// csrc/gemm_bf16x3.hip per wavefront = 48 MFMAs (2 x 2 accumulator tiles, 6 partial products, 2 k halves) fed by 24 ds_read_b128 operand
// fragments; then + the operand-split VALU (88 per step), + 12 ds_write_b128, + 2 workgroup barriers.  256-thread workgroups, 1 / 2 / 3 per
// CU (the LDS allocation sets the occupancy).  Reports bf16 TFLOP/s of the whole chip and the percentage of the register-only loop.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_bf16 mfma_lds_bf16.hip ; run: ./mfma_lds_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
union Frag { bf16x8 v; uint4 q; };
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// READS: 0 = operands stay in registers, 1 = 24 ds_read_b128 per step feeding the MFMAs; VALU: split fillers per step; WRITES: ds_write_b128 per
// step; BARS: barriers per step
template <int READS, int VALU, int WRITES, int BARS>
__global__ void __launch_bounds__(256, 2) probe(float *out, int iters, int lds_words)
{
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 6 * 516; i += 256) lds[i] = make_uint4(0x3f803f80u + i, 0x3f003f00u, 0x3f803f00u, 0x3f003f80u);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Frag a[2][3], b[2][3];
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 3; ++t) { a[i][t].q = lds[tid + 7 * i + t]; b[i][t].q = lds[tid + 300 + 5 * i + t]; }
    const int arow = (lane >> 5) * 129 + 64 * (wid >> 1) + (lane & 31), brow = (lane >> 5) * 129 + 64 * (wid & 1) + (lane & 31);
    float f[8]; unsigned g[8];
    for (int i = 0; i < 8; ++i) { g[i] = tid * 2654435761u + i; f[i] = 1.0f + tid * 1e-3f + i; }
    uint4 wv = make_uint4(tid, tid + 1, tid + 2, tid + 3);
    for (int it = 0; it < iters; ++it) {
        if (BARS >= 1) __syncthreads();
#pragma unroll
        for (int k = 0; k < VALU; ++k) {
            const int r = k & 7;
            if ((k % 3) == 0) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(g[r]) : "v"(f[(r + 1) & 7]));
            else if ((k % 3) == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[r]) : "v"(f[(r + 3) & 7]), "v"(g[(r + 5) & 7]));
            else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(g[r]) : "v"(g[(r + 2) & 7]), "v"(f[(r + 4) & 7]), "s"(0x07060302u));
        }
#pragma unroll
        for (int k = 0; k < WRITES; ++k) { wv.x = g[k & 7]; lds[6 * 516 + (k * 256 + tid) % 3096] = wv; }
        if (BARS >= 2) __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (READS) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        a[i][t].q = lds[t * 516 + 2 * ks * 129 + arow + 32 * i];
                        b[i][t].q = lds[(3 + t) * 516 + 2 * ks * 129 + brow + 32 * i];
                    }
            }
#define PROD(ta, tb) do { \
            acc[0][0] = MFMA(a[0][ta].v, b[0][tb].v, acc[0][0]); acc[0][1] = MFMA(a[0][ta].v, b[1][tb].v, acc[0][1]); \
            acc[1][0] = MFMA(a[1][ta].v, b[0][tb].v, acc[1][0]); acc[1][1] = MFMA(a[1][ta].v, b[1][tb].v, acc[1][1]); } while (0)
            PROD(1, 1); PROD(0, 2); PROD(2, 0); PROD(0, 1); PROD(1, 0); PROD(0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)g[i];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

static double base_tf[4];
template <int READS, int VALU, int WRITES, int BARS>
void run(int per_cu, const char *what)
{
    const int iters = 400, nb = 256 * per_cu;
    const int lds_bytes = per_cu == 1 ? 120 * 1024 : per_cu == 2 ? 72 * 1024 : 50 * 1024;     // 160 KB per CU: sets the occupancy
    float *out;
    hipMalloc(&out, nb * 256 * 4);
    hipFuncSetAttribute((const void *)probe<READS, VALU, WRITES, BARS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<READS, VALU, WRITES, BARS><<<nb, 256, lds_bytes>>>(out, 10, 0);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<READS, VALU, WRITES, BARS><<<nb, 256, lds_bytes>>>(out, iters, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = (double)nb * 4 * iters * 48 * 32768.0 / (best * 1e-3) / 1e12;
    if (!READS && !VALU && !WRITES && !BARS) base_tf[per_cu] = tf;
    printf("{\"what\": \"%s\", \"ds_read_b128\": %d, \"valu\": %d, \"ds_write_b128\": %d, \"barriers\": %d, \"workgroups_per_cu\": %d, \"ms\": %.3f, \"bf16_tflops\": %.0f, \"of_register_only\": %.3f}\n",
           what, READS ? 24 : 0, VALU, WRITES, BARS, per_cu, best, tf, tf / base_tf[per_cu]);
    fflush(stdout);
    hipFree(out);
}

int main()
{
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        run<0, 0, 0, 0>(per_cu, "48 MFMA per step, operands in registers");
        run<1, 0, 0, 0>(per_cu, "+ 24 ds_read_b128 fragments");
        run<0, 88, 0, 0>(per_cu, "registers + 88 VALU");
        run<1, 88, 0, 0>(per_cu, "reads + 88 VALU");
        run<1, 88, 12, 0>(per_cu, "reads + VALU + 12 ds_write_b128");
        run<1, 88, 12, 2>(per_cu, "reads + VALU + writes + 2 barriers (the gemm_bf16x3 step)");
        run<1, 0, 12, 2>(per_cu, "reads + writes + 2 barriers, no VALU");
        run<1, 88, 0, 2>(per_cu, "reads + VALU + 2 barriers, no writes");
        run<0, 88, 12, 2>(per_cu, "VALU + writes + 2 barriers, operands in registers");
    }
    return 0;
}
