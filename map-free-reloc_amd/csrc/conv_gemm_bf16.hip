// conv_gemm_bf16.hip -- the 3x3 decoder convolutions of the relative-pose regression encoder as IMPLICIT GEMMs on
// the gfx950 bf16 matrix cores: forward, gradient with respect to the input and gradient with respect to the weights are the same
// kernel, a "segmented" NT product
//
//        C[i, j] = sum_k  A[i, k] * B[j, k]            bf16 operands, fp32 accumulate, bf16 or fp32 result
//
// whose k axis is cut into segments of Lk elements, each with its own base offset inside A and inside B (two small tables).
//
// Reference call site: lib/models/regression/encoder/resunet.py:16-38, 75-128 (`conv` = Conv2d(3x3, padding 1) + BatchNorm + ELU inside
// `upconv4 / iconv4 / upconv3 / iconv3`: 1024->512 and 1024->512 channels at 46x34, 512->256 and 512->256 at 92x68 for the Map-free 360x270
// input) under Lightning's bf16 autocast (config/regression/mapfree/*.yaml + train.py:20-70).  Rounds 1-2 ran them through
// MIOpen (180 TFLOP/s = 7 % of the bf16 peak, 40 % of the training step).
//
// How a convolution becomes this product (host side: map-free-reloc_amd/regression/conv_bf16.py):
//   forward   rows i = pixels of the zero-haloed NHWC image [B, H+2, W+2, Cin], flattened; a filter tap (ky, kx) is a CONSTANT row
//             shift ((ky-1)(W+2) + kx-1) of that matrix, so segment = tap, Lk = Cin, A offset = shift * Cin, and B = the weights as
//             [Cout][tap][Cin].  Outputs are produced for every haloed position (5-10 % more rows than pixels) and the caller reads
//             the interior -- in exchange the operand addresses are affine: no bounds test, no im2col buffer.
//   d input   the same with the haloed NHWC output gradient as A and the 180-degree rotated, transposed weights as B.
//   d weight  rows i = input channels, rows j = output channels, k = pixels: both operands in zero-haloed channel-major images
//             [B][C][L] (row stride padded to 8 pixels so that the ky shift keeps 16-byte alignment; the kx shift is applied when the three
//             shifted copies of the input are written); segment = image, Lk = L; one grid.z slice per tap x K split, fp32 partials,
//             summed in a fixed order by the caller (no atomics: bit-reproducible).
//
// Mapping: workgroup = 256 A rows x 128 B rows, 4 wavefronts as 2 x 2, each 128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128
// accumulator registers), K in steps of 32.  The MFMA's first operand is the B row block, so a lane ends up with 4 CONSECUTIVE j for one
// i: 8-byte (bf16) or 16-byte (fp32) stores.  Operands are staged global -> registers -> LDS in fragment order ([k group of 8][row][16 B],
// rows padded by one unit: conflict-free ds_write_b128 and ds_read_b128), two LDS stages, ONE barrier per K step: the loads of step k+2
// are in flight and the LDS image of step k+1 is written while step k is multiplied.  Two workgroups per CU.
// Bound: bf16 MFMA (2 * 9 * Cin * Cout flops per output pixel against ~2 (Cin + Cout) bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CG_BM 256
#define CG_BN 128
#define CG_BK 32
#define CG_A_KG (CG_BM + 1)                     // 16-byte units per k group of the A image
#define CG_B_KG (CG_BN + 1)
#define CG_STAGE_UNITS (4 * CG_A_KG + 4 * CG_B_KG)

union CgFrag { bf16x8 v; uint4 q; };

__device__ __forceinline__ unsigned cg_bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;          // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

#define CG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <bool OUT_BF16>
__global__ void __launch_bounds__(256, 2)
conv_gemm_bf16_kernel(const unsigned short *__restrict__ A, long long sA, const long long *__restrict__ segA,
                      const unsigned short *__restrict__ B, long long sB, const long long *__restrict__ segB,
                      int Lk, int nkc_total, int nkc_z, const float *__restrict__ bias, void *__restrict__ Cout, long long ldc,
                      int M, int N, int nnb,
                      const long long *__restrict__ zA, const long long *__restrict__ zB, const long long *__restrict__ zC, const int *__restrict__ zk)
{
    __shared__ uint4 lds[2 * CG_STAGE_UNITS];            // 2 x 24.7 KB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid >> 1, wj = wid & 1;
    const int nb = blockIdx.x % nnb, mb = blockIdx.x / nnb;      // j blocks innermost: workgroups sharing an A row block run back to back
    const int z = blockIdx.z;
    const int i0 = mb * CG_BM, j0 = nb * CG_BN;
    const int kc0 = zk ? zk[z] : 0;
    const int kc1 = min(kc0 + nkc_z, nkc_total);
    const int nkb = kc1 - kc0;
    A += zA ? zA[z] : 0;
    B += zB ? zB[z] : 0;

    // staging assignment: A unit u = tid + 256 t (t = 0..3) -> (row u >> 2, k group u & 3); B unit u = tid + 256 t (t = 0..1)
    const unsigned short *ap[4], *bp[2];
    int adst[4], bdst[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int u = tid + 256 * t, row = u >> 2, kg = u & 3;
        ap[t] = A + (long long)min(i0 + row, M - 1) * sA + 8 * kg;           // rows beyond M: a valid row is read, its results are never stored
        adst[t] = kg * CG_A_KG + row;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = tid + 256 * t, row = u >> 2, kg = u & 3;
        bp[t] = B + (long long)min(j0 + row, N - 1) * sB + 8 * kg;
        bdst[t] = 4 * CG_A_KG + kg * CG_B_KG + row;
    }

    // k position of the next chunk to LOAD: segment index + offset inside it (wave-uniform; no division in the loop)
    const int cps = Lk / CG_BK;                                               // chunks per segment
    int seg = kc0 / cps, within = (kc0 - seg * cps) * CG_BK;
    long long offA = segA[seg] + within, offB = segB[seg] + within;

    // Three register sets (P, Q, R) rotate through the K steps: a chunk is loaded THREE steps before it is multiplied (global latency under
    // load is ~2 us, a K step of two co-resident workgroups ~0.45 us: with one step of distance the kernel ran at the speed of the memory
    // latency, 23 % of the matrix peak).  Named registers + macros: arrays indexed by the step parity end up in scratch memory.
    uint4 pa0, pa1, pa2, pa3, pb0, pb1, qa0, qa1, qa2, qa3, qb0, qb1, sa0, sa1, sa2, sa3, sb0, sb1;
    int issued = 0;                                        // chunks requested so far; past the last chunk the same (valid) addresses are re-read
#define CG_ADVANCE() do { if (++issued < nkb) { within += CG_BK; \
        if (within == Lk) { within = 0; ++seg; offA = segA[seg]; offB = segB[seg]; } else { offA += CG_BK; offB += CG_BK; } } } while (0)
#define CG_GLOAD(x) do { \
        x##a0 = *(const uint4 *)(ap[0] + offA); x##a1 = *(const uint4 *)(ap[1] + offA); x##a2 = *(const uint4 *)(ap[2] + offA); x##a3 = *(const uint4 *)(ap[3] + offA); \
        x##b0 = *(const uint4 *)(bp[0] + offB); x##b1 = *(const uint4 *)(bp[1] + offB); CG_ADVANCE(); } while (0)
#define CG_LSTORE(x, st) do { uint4 *l_ = lds + (st) * CG_STAGE_UNITS; \
        l_[adst[0]] = x##a0; l_[adst[1]] = x##a1; l_[adst[2]] = x##a2; l_[adst[3]] = x##a3; l_[bdst[0]] = x##b0; l_[bdst[1]] = x##b1; } while (0)

    f32x16 acc[2][4];                                      // [j tile][i tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses: operand row = tile base + (lane & 31), k group = 2 ks + (lane >> 5)
    const int arow = (lane >> 5) * CG_A_KG + 128 * wi + (lane & 31);
    const int brow = 4 * CG_A_KG + (lane >> 5) * CG_B_KG + 64 * wj + (lane & 31);

#define CG_COMPUTE(st) do { const uint4 *cur = lds + (st) * CG_STAGE_UNITS; \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
            CgFrag fa[4], fb[2]; \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) fa[t].q = cur[2 * ks * CG_A_KG + arow + 32 * t]; \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) fb[t].q = cur[2 * ks * CG_B_KG + brow + 32 * t]; \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) { \
                acc[0][t] = CG_MFMA(fb[0].v, fa[t].v, acc[0][t]); \
                acc[1][t] = CG_MFMA(fb[1].v, fa[t].v, acc[1][t]); } } } while (0)
    // one K step: the chunk three steps ahead is requested into the set that was just written out, the current stage is multiplied,
    // then (MFMAs queued, the wave can wait for memory) the chunk of the NEXT step goes to the other LDS stage; one barrier.
    // Inside the main loop every load and every LDS store is UNCONDITIONAL (past the end the last chunk is re-read and a stage nobody
    // reads is rewritten): with a vector-memory instruction under a condition the compiler's wait-count bookkeeping gives up at the
    // loop header and drains the whole queue (s_waitcnt vmcnt(0)) every third step -- the prefetch distance collapses to one step.
#define CG_STEP(kb, ld, st) do { \
        CG_GLOAD(ld); \
        CG_COMPUTE((kb) & 1); \
        CG_LSTORE(st, ((kb) + 1) & 1); \
        __syncthreads(); } while (0)

    if (nkb > 0) {
        CG_GLOAD(p);                                        // chunk 0
        CG_GLOAD(q);                                        // chunk 1 (or 0 again)
        CG_GLOAD(s);                                        // chunk 2
        CG_LSTORE(p, 0);
        __syncthreads();
        // step kb multiplies chunk kb, stores chunk kb+1 (sets q, s, p, q, ...) and loads chunk kb+3 into the set chunk kb used (p, q, s, ...)
        int kb = 0;
        for (; kb + 2 < nkb; kb += 3) {
            CG_STEP(kb, p, q);
            CG_STEP(kb + 1, q, s);
            CG_STEP(kb + 2, s, p);
        }
        if (kb < nkb) {                                     // one or two steps left; sets p, q hold chunks kb, kb+1 (q is already in the other stage? no: stored below)
            CG_COMPUTE(kb & 1);
            if (kb + 1 < nkb) {
                CG_LSTORE(q, (kb + 1) & 1);
                __syncthreads();
                CG_COMPUTE((kb + 1) & 1);
            }
        }
    }

    // epilogue: accumulator register r of tile (a, b): j = j0 + 64 wj + 32 a + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), i = i0 + 128 wi + 32 b + (lane & 31)
    const long long cbase = zC ? zC[z] : 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int i = i0 + 128 * wi + 32 * b + (lane & 31);
        if (i >= M) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j = j0 + 64 * wj + 32 * a + 8 * g + 4 * (lane >> 5);
                if (j >= N) continue;                       // N % 4 == 0 (host check): a group of 4 is inside or outside as a whole
                float v0 = acc[a][b][4 * g + 0], v1 = acc[a][b][4 * g + 1], v2 = acc[a][b][4 * g + 2], v3 = acc[a][b][4 * g + 3];
                if (bias) { const float4 bb = *(const float4 *)(bias + j); v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w; }
                if (OUT_BF16) {
                    uint2 o;
                    o.x = cg_bf16_rne(v0) | (cg_bf16_rne(v1) << 16);
                    o.y = cg_bf16_rne(v2) | (cg_bf16_rne(v3) << 16);
                    *(uint2 *)((unsigned short *)Cout + cbase + (long long)i * ldc + j) = o;
                } else {
                    *(float4 *)((float *)Cout + cbase + (long long)i * ldc + j) = make_float4(v0, v1, v2, v3);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// operand images (memory-bound; every output element is written, halo zeros included: no separate clear)
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short cg_ld_bf16(const void *x, long long i, int f32)
{
    return f32 ? (unsigned short)cg_bf16_rne(((const float *)x)[i]) : ((const unsigned short *)x)[i];
}

// x [B, C, H, W] -> out [B, H+2, Wp, C] bf16 with a zero halo (Wp = W + 2: a zero column on either side; Wp = W + 1: ONE zero column in
// front of every row, which is also the right neighbour of the row before it); block = one haloed row piece of 64 positions x 64
// channels, transposed through LDS
__global__ void __launch_bounds__(256) cg_pack_nhwc_halo_kernel(const void *__restrict__ x, int f32, int C, int H, int W, int Wp, unsigned short *__restrict__ out)
{
    __shared__ unsigned short t[64][66];
    const int Hp = H + 2;
    const int x0 = blockIdx.x * 64, byp = blockIdx.y, c0 = blockIdx.z * 64;
    const int b = byp / Hp, yp = byp - b * Hp;
    const int tid = threadIdx.x;
    const bool row_in = yp >= 1 && yp <= H;
    {
        const int xl = tid & 63, xs = x0 + xl - 1;                      // source column of haloed column x0 + xl
        const bool ok = row_in && xs >= 0 && xs < W && x0 + xl < Wp;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int cl = (tid >> 6) + 4 * k, c = c0 + cl;
            t[cl][xl] = (ok && c < C) ? cg_ld_bf16(x, (((long long)b * C + c) * H + (yp - 1)) * W + xs, f32) : (unsigned short)0;
        }
    }
    __syncthreads();
    const int cg = tid & 7;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int px = (tid >> 3) + 32 * k;
        if (x0 + px >= Wp || c0 + 8 * cg >= C) continue;               // C % 8 == 0 (host check)
        uint4 o;
        o.x = t[8 * cg + 0][px] | ((unsigned)t[8 * cg + 1][px] << 16);
        o.y = t[8 * cg + 2][px] | ((unsigned)t[8 * cg + 3][px] << 16);
        o.z = t[8 * cg + 4][px] | ((unsigned)t[8 * cg + 5][px] << 16);
        o.w = t[8 * cg + 6][px] | ((unsigned)t[8 * cg + 7][px] << 16);
        *(uint4 *)(out + (((long long)b * Hp + yp) * Wp + x0 + px) * C + c0 + 8 * cg) = o;
    }
}

// haloed NHWC result [B, H+2, Wp, N] bf16 -> y [B, N, H, W] bf16 (the layout the rest of the network uses); block = 64 pixels of one image
// row x 64 channels, transposed through LDS: 16-byte reads along the channels, 128-byte row pieces written along x
__global__ void __launch_bounds__(256) cg_unpack_nchw_kernel(const unsigned short *__restrict__ h, int N, int H, int W, int Wp, unsigned short *__restrict__ y)
{
    __shared__ unsigned short t[64][66];                   // [channel][pixel]
    const int x0 = blockIdx.x * 64, by = blockIdx.y, n0 = blockIdx.z * 64;
    const int b = by / H, yy = by - b * H;
    const int tid = threadIdx.x, cg = tid & 7;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int px = (tid >> 3) + 32 * k;
        if (x0 + px < W && n0 + 8 * cg < N) {
            const uint4 v = *(const uint4 *)(h + (((long long)b * (H + 2) + yy + 1) * Wp + x0 + px + 1) * N + n0 + 8 * cg);
            t[8 * cg + 0][px] = (unsigned short)v.x; t[8 * cg + 1][px] = (unsigned short)(v.x >> 16);
            t[8 * cg + 2][px] = (unsigned short)v.y; t[8 * cg + 3][px] = (unsigned short)(v.y >> 16);
            t[8 * cg + 4][px] = (unsigned short)v.z; t[8 * cg + 5][px] = (unsigned short)(v.z >> 16);
            t[8 * cg + 6][px] = (unsigned short)v.w; t[8 * cg + 7][px] = (unsigned short)(v.w >> 16);
        }
    }
    __syncthreads();
    const int xl = tid & 63;
    if (x0 + xl >= W) return;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int cl = (tid >> 6) + 4 * k;
        if (n0 + cl < N) y[(((long long)b * N + n0 + cl) * H + yy) * W + x0 + xl] = t[cl][xl];
    }
}

// x [B*C, H, W] -> ncopies channel-major haloed images [B*C][L] (rows of Wq >= W + 2 positions, Wq % 8 == 0, L >= (H + 2) Wq), copy k
// shifted by sh0 + k positions along a row: copy[k][p] = haloed[p + sh0 + k]; each copy sits between `slack` zero elements.
// One thread per 8 consecutive output elements (16-byte stores).
__global__ void __launch_bounds__(256) cg_pack_cm_halo_kernel(const void *__restrict__ x, int f32, long long BC, int H, int W, int Wq, long long L, int sh0,
                                                              long long slack, long long units_per_copy, unsigned short *__restrict__ out)
{
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= units_per_copy) return;
    const int k = blockIdx.y;
    const long long q = 8 * u - slack;                                  // position inside the copy's image area
    unsigned short v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (q >= 0 && q < BC * L) {
        const long long bc = q / L;
        const int pos = (int)(q - bc * L);
        const int yp = pos / Wq, xq0 = pos - yp * Wq;
        if (yp >= 1 && yp <= H) {
            const int sh = sh0 + k;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int xs = xq0 + e + sh - 1;                        // source column
                if (xs >= 0 && xs < W) v[e] = cg_ld_bf16(x, (bc * H + (yp - 1)) * W + xs, f32);
            }
        }
    }
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16); o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    *(uint4 *)(out + (long long)k * 8 * units_per_copy + 8 * u) = o;
}

extern "C" {

int mfr_conv_gemm_bf16(const void *A, long long sA, const long long *segA, const void *B, long long sB, const long long *segB,
                       int Lk, int nkc_total, int nkc_z, const float *bias, void *C, long long ldc, int out_dtype,
                       int M, int N, int nz, const long long *zA, const long long *zB, const long long *zC, const int *zk, void *stream)
{
    if (!A || !B || !C || !segA || !segB || M <= 0 || N <= 0 || nz <= 0 || Lk <= 0 || (Lk % CG_BK) || nkc_total <= 0 || nkc_z <= 0 ||
        (sA & 7) || (sB & 7) || (N & 3) || (ldc & 3) || ldc < N || (out_dtype != 0 && out_dtype != 1) || (nz > 1 && !zk && nkc_z < nkc_total && !zA))
        return MFR_E_ARG;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || (bias && ((uintptr_t)bias & 15))) return MFR_E_ARG;
    const int nnb = (N + CG_BN - 1) / CG_BN, nmb = (M + CG_BM - 1) / CG_BM;
    const dim3 grid((unsigned)(nnb * nmb), 1, (unsigned)nz);
    if (out_dtype == 1)
        hipLaunchKernelGGL(conv_gemm_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)A, sA, segA,
                           (const unsigned short *)B, sB, segB, Lk, nkc_total, nkc_z, bias, C, ldc, M, N, nnb, zA, zB, zC, zk);
    else
        hipLaunchKernelGGL(conv_gemm_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short *)A, sA, segA,
                           (const unsigned short *)B, sB, segB, Lk, nkc_total, nkc_z, bias, C, ldc, M, N, nnb, zA, zB, zC, zk);
    CHECK_LAUNCH();
    return 0;
}


int mfr_conv_pack_nhwc_halo(const void *x, int x_dtype, int B, int C, int H, int W, int Wp, void *out, int guard_rows, void *stream)
{
    if (!x || !out || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0 || (Wp != W + 1 && Wp != W + 2) || guard_rows < 0 || (x_dtype != 0 && x_dtype != 1) || ((uintptr_t)out & 15)) return MFR_E_ARG;
    unsigned short *o = (unsigned short *)out;
    const long long Mp = (long long)B * (H + 2) * Wp;
    if (guard_rows) {
        if (mfr_zero_async(o, (size_t)guard_rows * C * 2, (hipStream_t)stream) != hipSuccess) return MFR_E_LAUNCH;
        if (mfr_zero_async(o + ((long long)guard_rows + Mp) * C, (size_t)guard_rows * C * 2, (hipStream_t)stream) != hipSuccess) return MFR_E_LAUNCH;
    }
    hipLaunchKernelGGL(cg_pack_nhwc_halo_kernel, dim3((unsigned)((Wp + 63) / 64), (unsigned)(B * (H + 2)), (unsigned)((C + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, x, x_dtype == 0 ? 1 : 0, C, H, W, Wp, o + (long long)guard_rows * C);
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv_unpack_nchw(const void *haloed, int B, int N, int H, int W, int Wp, void *y, void *stream)
{
    if (!haloed || !y || B <= 0 || N <= 0 || (N & 7) || H <= 0 || W <= 0 || (Wp != W + 1 && Wp != W + 2) || ((uintptr_t)haloed & 15)) return MFR_E_ARG;
    hipLaunchKernelGGL(cg_unpack_nchw_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)(B * H), (unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)haloed, N, H, W, Wp, (unsigned short *)y);
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv_pack_cm_halo(const void *x, int x_dtype, long long BC, int H, int W, int Wq, long long L, int ncopies, int first_shift, long long slack,
                          void *out, void *stream)
{
    if (!x || !out || BC <= 0 || H <= 0 || W <= 0 || Wq < W + 2 || (Wq & 7) || L < (long long)(H + 2) * Wq || (L & 7) || ncopies <= 0 || slack < 0 || (slack & 7) ||
        (x_dtype != 0 && x_dtype != 1) || ((uintptr_t)out & 15)) return MFR_E_ARG;
    const long long units = (2 * slack + BC * L) / 8;
    hipLaunchKernelGGL(cg_pack_cm_halo_kernel, dim3((unsigned)((units + 255) / 256), (unsigned)ncopies), dim3(256), 0, (hipStream_t)stream,
                       x, x_dtype == 0 ? 1 : 0, BC, H, W, Wq, L, first_shift, slack, units, (unsigned short *)out);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
