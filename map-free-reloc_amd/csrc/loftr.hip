// loftr.hip -- LoFTR matcher kernels on gfx950.
//
// Reference call site: LoFTR_matcher.match (etc/feature_matching_baselines/matchers.py:24-59) ->
// upstream LoFTR(default_cfg) (un-vendored submodule, .gitmodules:1-3; algorithm per SURVEY.md
// Appendix A.4).  Hand-written here:
//
//   la_kv_kernel / la_out_kernel   linear attention of the coarse transformer (d_model 256, 8 heads x 32,
//       phi = elu + 1):  KV = phi(K)^T (V / L) and K-sum reduced over the 6120 tokens on the exact
//       fp32 matrix cores (v_mfma_f32_32x32x2_f32, 2 tokens per instruction), then
//       out = phi(Q) KV / (phi(Q).Ksum + eps) * L, again on MFMA.  HBM-bound: Q, K, V are read once,
//       the message is written once; per-chunk partial KV are summed in a FIXED order (deterministic).
//   dsm_* kernels   dual-softmax coarse matching: conf = softmax_row(S) * softmax_col(S) with
//       S = f0 f1^T / (C * temperature) materialised ONCE (by the GEMM) and then only streamed:
//       row/col (max, sum-exp) statistics, row arg-max / col max of conf, threshold 0.2, border 2,
//       mutual-max check, ordered compaction -> (i, j, conf) per match.  Upstream materialises sim,
//       two softmaxes, conf, the mask and two max reductions (>= 6 full 150 MB tensors per pair).
//   fine_gather_kernel   the 5x5 windows of the 1/2-resolution feature map around every coarse match
//       (upstream unfolds the WHOLE map: 78 MB per image) gathered straight from NHWC features.
//
// Compiled with -ffp-contract=off: conf values are compared for equality across kernels (mutual-max
// test), so every kernel must evaluate the same expression with the same roundings.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ float phi(float x) { return x > 0.f ? x + 1.f : expf(x); }   // elu(x) + 1

#define LA_DH 32
#define LA_CHUNK 256          // tokens per workgroup in the KV pass (64 per wavefront)

// grid (nchunks, H, B), 256 threads.  partial KV [B,H,nchunks,32,32] (d-major), partial Ksum [B,H,nchunks,32]
__global__ void __launch_bounds__(256) la_kv_kernel(const float *__restrict__ K, const float *__restrict__ V, int ld, int L,
                                                    float inv_len, float *__restrict__ kv_part, float *__restrict__ ks_part)
{
    __shared__ float red[4][32 * 32];
    __shared__ float redk[4][64];
    const int c = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nchunks = gridDim.x, H = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane & 31, half = lane >> 5;
    const float *Kb = K + (size_t)b * L * ld + h * LA_DH + col;
    const float *Vb = V + (size_t)b * L * ld + h * LA_DH + col;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ks = 0.f;
    const int t0 = c * LA_CHUNK + wid * 64;
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
        const int tok = t0 + 2 * s + half;
        float kk = 0.f, vv = 0.f;
        if (tok < L) { kk = phi(Kb[(size_t)tok * ld]); vv = Vb[(size_t)tok * ld] * inv_len; }
        ks += kk;
        // A[i = d][k = token], B[k = token][j = v]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk, vv, acc, 0, 0, 0);
    }
    // C[row = d][col = v]: rows (r&3) + 8 (r>>2) + 4 half
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = acc[r];
    redk[wid][lane] = ks;
    __syncthreads();
    float *o = kv_part + (((size_t)b * H + h) * nchunks + c) * 1024;
    for (int i = tid; i < 1024; i += 256) o[i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    if (tid < 32) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s = s + (redk[w][tid] + redk[w][tid + 32]);
        ks_part[(((size_t)b * H + h) * nchunks + c) * 32 + tid] = s;
    }
}

// grid (H, B): the chunk partials of one (image, head) summed in chunk order INTO chunk 0's slot (round 5).  Rounds 1-4 left this sum to every
// wavefront of la_out_kernel: 19 chunks x 32 values per lane against 16 query values -- 95 % of that kernel's loads, 0.25 ms per launch for
// 63 us of HBM traffic.  (0 + p0) + p1 + ... there, p0 + p1 + ... here: the same sums, bit for bit.
__global__ void __launch_bounds__(256) la_fold_kernel(int nchunks, float *__restrict__ kv_part, float *__restrict__ ks_part)
{
    const int h = blockIdx.x, b = blockIdx.y, H = gridDim.x, tid = threadIdx.x;
    float *p = kv_part + ((size_t)b * H + h) * nchunks * 1024;
    float *pk = ks_part + ((size_t)b * H + h) * nchunks * 32;
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = p[tid + 256 * q];
    for (int c = 1; c < nchunks; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = acc[q] + p[(size_t)c * 1024 + tid + 256 * q];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[tid + 256 * q] = acc[q];
    if (tid < 32) {
        float a = pk[tid];
        for (int c = 1; c < nchunks; ++c) a = a + pk[(size_t)c * 32 + tid];
        pk[tid] = a;
    }
}

// grid (ceil(L/128), H, B), 256 threads: wavefront = 32 query tokens; nsum = 1: chunk 0's slot holds the folded sums (la_fold_kernel)
__global__ void __launch_bounds__(256) la_out_kernel(const float *__restrict__ Q, int ldq, int L, int nchunks, int nsum,
                                                     const float *__restrict__ kv_part, const float *__restrict__ ks_part,
                                                     float v_len, float *__restrict__ out, int ldo)
{
    const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane & 31, half = lane >> 5;
    // B operand: KV[d = 16 half + s][v = col], summed over the chunks in chunk order
    float kv[16], ksum[16];
    {
        const float *p = kv_part + ((size_t)b * H + h) * nchunks * 1024;
        const float *pk = ks_part + ((size_t)b * H + h) * nchunks * 32;
#pragma unroll
        for (int s = 0; s < 16; ++s) { kv[s] = 0.f; ksum[s] = 0.f; }
        for (int c = 0; c < nsum; ++c) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                kv[s] = kv[s] + p[(size_t)c * 1024 + (16 * half + s) * 32 + col];
                ksum[s] = ksum[s] + pk[(size_t)c * 32 + 16 * half + s];
            }
        }
    }
    const int tok = blockIdx.x * 128 + wid * 32 + col;
    float q[16];
    {
        const bool ok = tok < L;
        const float4 *qp = (const float4 *)(Q + ((size_t)b * L + (ok ? tok : 0)) * ldq + h * LA_DH + 16 * half);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = ok ? qp[g] : make_float4(0.f, 0.f, 0.f, 0.f);
            q[4 * g] = ok ? phi(t.x) : 0.f; q[4 * g + 1] = ok ? phi(t.y) : 0.f;
            q[4 * g + 2] = ok ? phi(t.z) : 0.f; q[4 * g + 3] = ok ? phi(t.w) : 0.f;
        }
    }
    float dz = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) dz = dz + q[s] * ksum[s];
    dz = dz + __shfl_xor(dz, 32, 64);
    const float z = 1.0f / (dz + 1e-6f);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q[s], kv[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float zr = __shfl(z, row, 64);
        const int t = blockIdx.x * 128 + wid * 32 + row;
        if (t < L) out[((size_t)b * L + t) * ldo + h * LA_DH + col] = (acc[r] * zr) * v_len;
    }
}

// ------------------------------------------------------------------------------------------ dual softmax
struct Lse { float m, s; };
static __device__ __forceinline__ void lse_add(Lse &a, float x)
{
    if (x > a.m) { a.s = a.s * expf(a.m - x) + 1.f; a.m = x; }
    else a.s = a.s + expf(x - a.m);
}
static __device__ __forceinline__ void lse_merge(Lse &a, float m, float s)
{
    if (m == -INFINITY) return;
    if (m > a.m) { a.s = a.s * expf(a.m - m) + s; a.m = m; }
    else a.s = a.s + s * expf(m - a.m);
}
// conf_ij = softmax over rows-dim * softmax over cols-dim; ONE definition used by every kernel
static __device__ __forceinline__ float conf_val(float s, float rmax, float rsum, float cmax, float csum)
{
    return (expf(s - cmax) / csum) * (expf(s - rmax) / rsum);
}

// row statistics (over j) : one wavefront per row
__global__ void __launch_bounds__(256) dsm_rowstat_kernel(const float *__restrict__ S, int L0, int L1, float temp,
                                                          float *__restrict__ rmax, float *__restrict__ rsum)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= L0) return;
    const float *row = S + ((size_t)b * L0 + i) * L1;
    Lse a = { -INFINITY, 0.f };
    for (int j = lane; j < L1; j += 64) lse_add(a, row[j] / temp);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float om = __shfl_xor(a.m, off, 64), os = __shfl_xor(a.s, off, 64);
        lse_merge(a, om, os);
    }
    if (lane == 0) { rmax[(size_t)b * L0 + i] = a.m; rsum[(size_t)b * L0 + i] = a.s; }
}

// column statistics (over i): 64 columns x 16 row groups
__global__ void __launch_bounds__(1024) dsm_colstat_kernel(const float *__restrict__ S, int L0, int L1, float temp,
                                                           float *__restrict__ cmax, float *__restrict__ csum)
{
    __shared__ float sm[16][64], ss[16][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6, j = blockIdx.x * 64 + lane;
    Lse a = { -INFINITY, 0.f };
    if (j < L1) {
        const float *colp = S + (size_t)b * L0 * L1 + j;
        for (int i = g; i < L0; i += 16) lse_add(a, colp[(size_t)i * L1] / temp);
    }
    sm[g][lane] = a.m; ss[g][lane] = a.s;
    __syncthreads();
    if (g == 0 && j < L1) {
        for (int k = 1; k < 16; ++k) lse_merge(a, sm[k][lane], ss[k][lane]);
        cmax[(size_t)b * L1 + j] = a.m; csum[(size_t)b * L1 + j] = a.s;
    }
}

__global__ void __launch_bounds__(256) dsm_rowbest_kernel(const float *__restrict__ S, int L0, int L1, float temp,
                                                          const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                          const float *__restrict__ cmax, const float *__restrict__ csum,
                                                          float *__restrict__ rbest, int *__restrict__ rarg)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= L0) return;
    const float *row = S + ((size_t)b * L0 + i) * L1;
    const float rm = rmax[(size_t)b * L0 + i], rs = rsum[(size_t)b * L0 + i];
    float best = -1.f; int bj = 0x7fffffff;
    for (int j = lane; j < L1; j += 64) {
        const float c = conf_val(row[j] / temp, rm, rs, cmax[(size_t)b * L1 + j], csum[(size_t)b * L1 + j]);
        if (c > best) { best = c; bj = j; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64); const int oj = __shfl_xor(bj, off, 64);
        if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
    }
    if (lane == 0) { rbest[(size_t)b * L0 + i] = best; rarg[(size_t)b * L0 + i] = bj; }
}

__global__ void __launch_bounds__(1024) dsm_colbest_kernel(const float *__restrict__ S, int L0, int L1, float temp,
                                                           const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                           const float *__restrict__ cmax, const float *__restrict__ csum,
                                                           float *__restrict__ cbest)
{
    __shared__ float sb[16][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6, j = blockIdx.x * 64 + lane;
    float best = -1.f;
    if (j < L1) {
        const float *colp = S + (size_t)b * L0 * L1 + j;
        const float cm = cmax[(size_t)b * L1 + j], cs = csum[(size_t)b * L1 + j];
        for (int i = g; i < L0; i += 16) {
            const float c = conf_val(colp[(size_t)i * L1] / temp, rmax[(size_t)b * L0 + i], rsum[(size_t)b * L0 + i], cm, cs);
            if (c > best) best = c;
        }
    }
    sb[g][lane] = best;
    __syncthreads();
    if (g == 0 && j < L1) {
        for (int k = 1; k < 16; ++k) if (sb[k][lane] > best) best = sb[k][lane];
        cbest[(size_t)b * L1 + j] = best;
    }
}

// ---- two-pass variant: ONE sweep over S for both logsumexps, ONE sweep for both arg-max searches ----------------
// A workgroup owns a tile of DSM_RS rows x DSM_CB columns (a thread owns 4 adjacent columns: 16-byte loads, a row of the
// tile is 4 KB contiguous).  Row quantities are reduced inside the tile (wave shuffles, then the 4 waves in column order)
// and written as one partial per (column block, row); column quantities are carried in registers down the tile's rows and
// written as one partial per (row stripe, column).  Two small kernels fold the partials in a fixed order (deterministic).
// HBM traffic: S is read twice (the four-pass variant reads it four times); partials add 3 % of S.
#define DSM_RS 64
#define DSM_CB 1024
#define DSM_RU 4

// Round 5: the two tile kernels in the arithmetic the Sinkhorn sweep got (csrc/superglue_match.hip) -- rounds 2-4 spent ~50 instructions per
// matrix element here (an IEEE division by the temperature, four precise expf in two online log-sum-exp updates, a butterfly of rescaling merges
// per row; the arg-max sweep two expf and two divisions): 1.65 + 1.20 ms per 16 pairs at 0.9 TB/s.  Now: s * (1 / temperature); a row's
// wavefront-wide maximum first and ONE v_exp_f32 per term; column partials over the four rows in flight at once (1.5 per term); the
// confidences from two v_exp_f32 and the reciprocals of the sums (per column: registers, per row: LDS).  The four-sweep kernels (variant 1)
// keep the round-1 arithmetic; the two agree to round-off (tests/test_gpu_loftr_parity.py).
template <int CTRL> static __device__ __forceinline__ float dsm_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ void dsm_swap16(float &a, float &b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
static __device__ __forceinline__ void dsm_swap32(float &a, float &b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
static __device__ __forceinline__ float dsm_wave_max(float x)
{
    x = fmaxf(x, dsm_dpp<0xB1>(x)); x = fmaxf(x, dsm_dpp<0x4E>(x)); x = fmaxf(x, dsm_dpp<0x141>(x)); x = fmaxf(x, dsm_dpp<0x140>(x));
    float y = x; dsm_swap16(x, y); x = fmaxf(x, y);
    y = x; dsm_swap32(x, y); x = fmaxf(x, y);
    return x;
}
static __device__ __forceinline__ float dsm_wave_sum(float x)
{
    x += dsm_dpp<0xB1>(x); x += dsm_dpp<0x4E>(x); x += dsm_dpp<0x141>(x); x += dsm_dpp<0x140>(x);
    float y = x; dsm_swap16(x, y); x += y;
    y = x; dsm_swap32(x, y); x += y;
    return x;
}

static __device__ __forceinline__ void dsm_load4(const float *__restrict__ row, int j0, int nval, bool vec, float x[4])
{
    if (nval == 4 && vec) {
        const float4 t = *(const float4 *)(row + j0);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = k < nval ? row[j0 + k] : 0.f;
    }
}

__global__ void __launch_bounds__(256) dsm_stats_tile_kernel(const float *__restrict__ S, int L0, int L1, float temp,
                                                             float *__restrict__ rpm, float *__restrict__ rps,
                                                             float *__restrict__ cpm, float *__restrict__ cps)
{
    __shared__ float wm[DSM_RS][4], wsum[DSM_RS][4];
    const int st = blockIdx.x, cb = blockIdx.y, b = blockIdx.z, nst = gridDim.x, ncb = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j0 = cb * DSM_CB + 4 * tid, nval = min(4, max(0, L1 - j0));
    const int i0 = st * DSM_RS, nrows = min(DSM_RS, L0 - i0);
    const bool vec = (L1 & 3) == 0;
    const float *base = S + ((size_t)b * L0 + i0) * L1;
    Lse c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = { -INFINITY, 0.f };
    const float itemp = 1.0f / temp;
    for (int r0 = 0; r0 < nrows; r0 += DSM_RU) {
        float x[DSM_RU][4];
#pragma unroll
        for (int u = 0; u < DSM_RU; ++u)
            if (r0 + u < nrows) dsm_load4(base + (size_t)(r0 + u) * L1, j0, nval, vec, x[u]);
        float t[DSM_RU][4];
#pragma unroll
        for (int u = 0; u < DSM_RU; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[u][k] = (r0 + u < nrows && k < nval) ? x[u][k] * itemp : -INFINITY;
        // rows: the wavefront's maximum first, one exponential per term (an all-absent lane / row contributes exp(-inf) = 0)
#pragma unroll
        for (int u = 0; u < DSM_RU; ++u) {
            const float mx = dsm_wave_max(fmaxf(fmaxf(t[u][0], t[u][1]), fmaxf(t[u][2], t[u][3])));
            const float ms = (mx > -INFINITY) ? mx : 0.f;
            const float sm = dsm_wave_sum(((__expf(t[u][0] - ms) + __expf(t[u][1] - ms)) + __expf(t[u][2] - ms)) + __expf(t[u][3] - ms));
            if (lane == 0 && r0 + u < nrows) { wm[r0 + u][wid] = mx; wsum[r0 + u][wid] = sm; }
        }
        // columns: the four rows at once (row r0 exists; absent rows are -inf)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nval) {
                const float bm = fmaxf(fmaxf(t[0][k], t[1][k]), fmaxf(t[2][k], t[3][k]));
                const float bs = ((__expf(t[0][k] - bm) + __expf(t[1][k] - bm)) + __expf(t[2][k] - bm)) + __expf(t[3][k] - bm);
                const float nm = fmaxf(c[k].m, bm);
                c[k].s = c[k].s * __expf(c[k].m - nm) + bs * __expf(bm - nm);
                c[k].m = nm;
            }
    }
    __syncthreads();
    if (tid < nrows) {
        Lse a = { wm[tid][0], wsum[tid][0] };
#pragma unroll
        for (int w = 1; w < 4; ++w) lse_merge(a, wm[tid][w], wsum[tid][w]);
        const size_t o = ((size_t)b * ncb + cb) * L0 + i0 + tid;
        rpm[o] = a.m; rps[o] = a.s;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nval) {
            const size_t o = ((size_t)b * nst + st) * L1 + j0 + k;
            cpm[o] = c[k].m; cps[o] = c[k].s;
        }
}

// fold the partials: rows over the column blocks (ascending), columns over the row stripes (ascending)
__global__ void __launch_bounds__(256) dsm_stats_fold_kernel(int L0, int L1, int nst, int ncb, const float *__restrict__ rpm,
                                                             const float *__restrict__ rps, const float *__restrict__ cpm,
                                                             const float *__restrict__ cps, float *__restrict__ rmax,
                                                             float *__restrict__ rsum, float *__restrict__ cmax,
                                                             float *__restrict__ csum)
{
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t < L0) {
        Lse a = { -INFINITY, 0.f };
        for (int k = 0; k < ncb; ++k) lse_merge(a, rpm[((size_t)b * ncb + k) * L0 + t], rps[((size_t)b * ncb + k) * L0 + t]);
        rmax[(size_t)b * L0 + t] = a.m; rsum[(size_t)b * L0 + t] = a.s;
    }
    if (t < L1) {
        Lse a = { -INFINITY, 0.f };
        for (int k = 0; k < nst; ++k) lse_merge(a, cpm[((size_t)b * nst + k) * L1 + t], cps[((size_t)b * nst + k) * L1 + t]);
        cmax[(size_t)b * L1 + t] = a.m; csum[(size_t)b * L1 + t] = a.s;
    }
}

__global__ void __launch_bounds__(256) dsm_best_tile_kernel(const float *__restrict__ S, int L0, int L1, float temp,
                                                            const float *__restrict__ rmax, const float *__restrict__ rsum,
                                                            const float *__restrict__ cmax, const float *__restrict__ csum,
                                                            float *__restrict__ rbp, int *__restrict__ rap,
                                                            float *__restrict__ cbp)
{
    __shared__ float rm_s[DSM_RS], rs_s[DSM_RS], wb[DSM_RS][4];
    __shared__ int wj[DSM_RS][4];
    const int st = blockIdx.x, cb = blockIdx.y, b = blockIdx.z, nst = gridDim.x, ncb = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j0 = cb * DSM_CB + 4 * tid, nval = min(4, max(0, L1 - j0));
    const int i0 = st * DSM_RS, nrows = min(DSM_RS, L0 - i0);
    const bool vec = (L1 & 3) == 0;
    const float *base = S + ((size_t)b * L0 + i0) * L1;
    if (tid < nrows) { rm_s[tid] = rmax[(size_t)b * L0 + i0 + tid]; rs_s[tid] = 1.0f / rsum[(size_t)b * L0 + i0 + tid]; }      // (reciprocal of the row sum)
    const float itemp = 1.0f / temp;
    float cm[4], cs[4], cbest[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cm[k] = k < nval ? cmax[(size_t)b * L1 + j0 + k] : 0.f;
        cs[k] = k < nval ? 1.0f / csum[(size_t)b * L1 + j0 + k] : 1.f;                                                              // (reciprocal of the column sum)
        cbest[k] = -1.f;
    }
    __syncthreads();
    for (int r0 = 0; r0 < nrows; r0 += DSM_RU) {
        float x[DSM_RU][4];
#pragma unroll
        for (int u = 0; u < DSM_RU; ++u)
            if (r0 + u < nrows) dsm_load4(base + (size_t)(r0 + u) * L1, j0, nval, vec, x[u]);
        float best[DSM_RU]; int bj[DSM_RU];
#pragma unroll
        for (int u = 0; u < DSM_RU; ++u) {
            best[u] = -1.f; bj[u] = 0x7fffffff;
            if (r0 + u < nrows) {
                const float rm = rm_s[r0 + u], rs = rs_s[r0 + u];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nval) {
                        const float tt = x[u][k] * itemp;
                        const float cf = (__expf(tt - cm[k]) * cs[k]) * (__expf(tt - rm) * rs);      // conf_val with v_exp_f32 and the reciprocal sums
                        if (cf > best[u]) { best[u] = cf; bj[u] = j0 + k; }
                        if (cf > cbest[k]) cbest[k] = cf;
                    }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int u = 0; u < DSM_RU; ++u) {
                const float ob = __shfl_xor(best[u], off, 64); const int oj = __shfl_xor(bj[u], off, 64);
                if (ob > best[u] || (ob == best[u] && oj < bj[u])) { best[u] = ob; bj[u] = oj; }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < DSM_RU; ++u)
                if (r0 + u < nrows) { wb[r0 + u][wid] = best[u]; wj[r0 + u][wid] = bj[u]; }
        }
    }
    __syncthreads();
    if (tid < nrows) {
        float best = wb[tid][0]; int bj = wj[tid][0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wb[tid][w] > best || (wb[tid][w] == best && wj[tid][w] < bj)) { best = wb[tid][w]; bj = wj[tid][w]; }
        const size_t o = ((size_t)b * ncb + cb) * L0 + i0 + tid;
        rbp[o] = best; rap[o] = bj;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nval) cbp[((size_t)b * nst + st) * L1 + j0 + k] = cbest[k];
}

__global__ void __launch_bounds__(256) dsm_best_fold_kernel(int L0, int L1, int nst, int ncb, const float *__restrict__ rbp,
                                                            const int *__restrict__ rap, const float *__restrict__ cbp,
                                                            float *__restrict__ rbest, int *__restrict__ rarg,
                                                            float *__restrict__ cbest)
{
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t < L0) {
        float best = -1.f; int bj = 0x7fffffff;
        for (int k = 0; k < ncb; ++k) {
            const float v = rbp[((size_t)b * ncb + k) * L0 + t]; const int j = rap[((size_t)b * ncb + k) * L0 + t];
            if (v > best || (v == best && j < bj)) { best = v; bj = j; }
        }
        rbest[(size_t)b * L0 + t] = best; rarg[(size_t)b * L0 + t] = bj;
    }
    if (t < L1) {
        float best = -1.f;
        for (int k = 0; k < nst; ++k) best = fmaxf(best, cbp[((size_t)b * nst + k) * L1 + t]);
        cbest[(size_t)b * L1 + t] = best;
    }
}

// one workgroup per pair: threshold, border, mutual max, ordered compaction (upstream get_coarse_match)
__global__ void __launch_bounds__(256) dsm_match_kernel(int L0, int L1, int h0, int w0, int h1, int w1, float thr, int border,
                                                        const float *__restrict__ rbest, const int *__restrict__ rarg,
                                                        const float *__restrict__ cbest, int *__restrict__ i_ids,
                                                        int *__restrict__ j_ids, float *__restrict__ mconf,
                                                        int *__restrict__ n_match)
{
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < L0; start += 256) {
        const int i = start + tid;
        bool valid = false; int j = 0; float c = 0.f;
        if (i < L0) {
            c = rbest[(size_t)b * L0 + i]; j = rarg[(size_t)b * L0 + i];
            const int y0 = i / w0, x0 = i - y0 * w0, y1 = j / w1, x1 = j - y1 * w1;
            const bool inb = y0 >= border && y0 < h0 - border && x0 >= border && x0 < w0 - border &&
                             y1 >= border && y1 < h1 - border && x1 >= border && x1 < w1 - border;
            valid = (c > thr) && inb && (c == cbest[(size_t)b * L1 + j]);
        }
        const unsigned long long bal = __ballot(valid);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wid; ++w) off += wave_cnt[w];
        if (valid) {
            const int o = off + wpre;
            i_ids[(size_t)b * L0 + o] = i; j_ids[(size_t)b * L0 + o] = j; mconf[(size_t)b * L0 + o] = c;
        }
        __syncthreads();
        if (tid == 0) base_s = off + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_match[b] = base_s;
}

// ------------------------------------------------------------------------------------------ fine windows
// feat [Bimg, Hf, Wf, C] NHWC; one workgroup per match: 25 cells x C channels, zero padding
__global__ void __launch_bounds__(256) fine_gather_kernel(const float *__restrict__ feat, int Hf, int Wf, int C,
                                                          const int *__restrict__ img_ids, const int *__restrict__ cell_ids,
                                                          int wc, int stride, int win, float *__restrict__ out)
{
    const int m = blockIdx.x;
    const int img = img_ids[m], cell = cell_ids[m];
    const int cy = (cell / wc) * stride - win / 2, cx = (cell % wc) * stride - win / 2;
    const int total = win * win * C;
    if (!(C & 3) && !(((size_t)feat | (size_t)out) & 15)) {
        // 16-byte pieces (round 5; a window is win^2 runs of C contiguous floats on both sides): one division per PIECE by the piece count of a
        // run instead of two per element -- the element loop below spent most of its time on them (0.44 -> 0.15 ms per launch)
        const int c4 = C >> 2, total4 = win * win * c4;
        const float4 *f4 = (const float4 *)feat;
        float4 *o4 = (float4 *)out + (size_t)m * total4;
        for (int e = threadIdx.x; e < total4; e += 256) {
            const int k = e / c4, q = e - k * c4;
            const int ky = k / win, y = cy + ky, x = cx + (k - ky * win);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = f4[(((size_t)img * Hf + y) * Wf + x) * c4 + q];
            o4[e] = v;
        }
        return;
    }
    for (int e = threadIdx.x; e < total; e += 256) {
        const int k = e / C, ch = e - k * C;
        const int y = cy + k / win, x = cx + k % win;
        float v = 0.f;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = feat[(((size_t)img * Hf + y) * Wf + x) * C + ch];
        out[(size_t)m * total + e] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fine-level linear attention (LoFTR fine transformer: d_model 128, 8 heads x 16, L = W*W = 25 tokens per 5x5 window,
// tens of thousands of windows).  Upstream LinearAttention per window and head:
//     Q = elu(q)+1, K = elu(k)+1, V = v / L;  KV = K^T V (16x16);  Z = 1 / (Q . sum_s K + 1e-6);  out = (Q KV) Z L
// Through the library this is three batched einsums over 16x16 / 25x16 matrices (<5 % of peak).  Here one wavefront owns
// one window: lane (hh, j) owns channels c = 16 hh + j and 64 + c (heads hh and hh + 4); K stays in registers (own
// channel only), V and later Q are broadcast inside the 16-lane head group through LDS, KV[h][d][:] accumulates in 16
// registers per owned channel.  HBM traffic = q, k, v read once + message written once (51 KB per window).
#define FA_L 25
#define FA_D 128
__global__ void __launch_bounds__(128) fine_attention_kernel(const float *__restrict__ Q, int ldq, const float *__restrict__ K,
                                                             const float *__restrict__ V, int ld, int Bw, float *__restrict__ O, int ldo)
{
    __shared__ __attribute__((aligned(16))) float Xs[2][FA_L][FA_D];     // V, then phi(Q)
    __shared__ __attribute__((aligned(16))) float KVs[2][FA_D][16];      // [channel (h, d)][v]
    __shared__ float Kss[2][FA_D];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int win_raw = blockIdx.x * 2 + wv;
    const bool live = win_raw < Bw;
    const int win = live ? win_raw : Bw - 1;
    const int hh = lane >> 4;
    const float inv_len = 1.0f / FA_L;
    const float *kb = K + (size_t)win * FA_L * ld, *vb = V + (size_t)win * FA_L * ld, *qb = Q + (size_t)win * FA_L * ldq;

    float kq[2][FA_L], qr0[FA_L], qr1[FA_L];
    // round 6: the window's Q rows are requested together with K and V (150 loads in flight per lane) -- the kernel moves 51 KB per window at a few
    // FMAs per byte, and loading Q only after the K^T V phase exposed the HBM latency a second time with nothing else of this wavefront in flight
#pragma unroll
    for (int s = 0; s < FA_L; ++s) {
        kq[0][s] = kb[(size_t)s * ld + lane];
        kq[1][s] = kb[(size_t)s * ld + 64 + lane];
    }
    float vr0[FA_L], vr1[FA_L];
#pragma unroll
    for (int s = 0; s < FA_L; ++s) { vr0[s] = vb[(size_t)s * ld + lane]; vr1[s] = vb[(size_t)s * ld + 64 + lane]; }
#pragma unroll
    for (int s = 0; s < FA_L; ++s) { qr0[s] = qb[(size_t)s * ldq + lane]; qr1[s] = qb[(size_t)s * ldq + 64 + lane]; }
#pragma unroll
    for (int s = 0; s < FA_L; ++s) {
        kq[0][s] = phi(kq[0][s]); kq[1][s] = phi(kq[1][s]);
        Xs[wv][s][lane] = vr0[s] * inv_len;
        Xs[wv][s][64 + lane] = vr1[s] * inv_len;
    }
    __syncthreads();
    // KV[h][d = this lane's channel][v] = sum_s K[s][h,d] V[s][h,v]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float kv[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) kv[v] = 0.f;
        float ks = 0.f;
        const int hb = (hh + 4 * t) * 16;
#pragma unroll
        for (int s = 0; s < FA_L; ++s) {
            const float kk = kq[t][s];
            ks += kk;
            const float4 *vr = (const float4 *)&Xs[wv][s][hb];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 vv = vr[g];
                kv[4 * g] = fmaf(kk, vv.x, kv[4 * g]); kv[4 * g + 1] = fmaf(kk, vv.y, kv[4 * g + 1]);
                kv[4 * g + 2] = fmaf(kk, vv.z, kv[4 * g + 2]); kv[4 * g + 3] = fmaf(kk, vv.w, kv[4 * g + 3]);
            }
        }
        float4 *dst = (float4 *)KVs[wv][64 * t + lane];
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = make_float4(kv[4 * g], kv[4 * g + 1], kv[4 * g + 2], kv[4 * g + 3]);
        Kss[wv][64 * t + lane] = ks;
    }
    __syncthreads();                                        // V fully consumed, KV / Ksum visible
#pragma unroll
    for (int s = 0; s < FA_L; ++s) {
        Xs[wv][s][lane] = phi(qr0[s]);
        Xs[wv][s][64 + lane] = phi(qr1[s]);
    }
    __syncthreads();
    // out[l][h, v = this lane] = (sum_d Q[l][h,d] KV[h][d][v]) / (sum_d Q[l][h,d] Ksum[h,d] + eps) * L
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int hb = (hh + 4 * t) * 16, j = lane & 15;
        float kvc[16], ksr[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) { kvc[d] = KVs[wv][hb + d][j]; ksr[d] = Kss[wv][hb + d]; }
#pragma unroll 5
        for (int l = 0; l < FA_L; ++l) {
            const float4 *qr = (const float4 *)&Xs[wv][l][hb];
            float acc = 0.f, z = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 qq = qr[g];
                acc = fmaf(qq.x, kvc[4 * g], acc); acc = fmaf(qq.y, kvc[4 * g + 1], acc);
                acc = fmaf(qq.z, kvc[4 * g + 2], acc); acc = fmaf(qq.w, kvc[4 * g + 3], acc);
                z = fmaf(qq.x, ksr[4 * g], z); z = fmaf(qq.y, ksr[4 * g + 1], z);
                z = fmaf(qq.z, ksr[4 * g + 2], z); z = fmaf(qq.w, ksr[4 * g + 3], z);
            }
            if (live) O[((size_t)win * FA_L + l) * ldo + 64 * t + lane] = acc * (1.0f / (z + 1e-6f)) * (float)FA_L;
        }
    }
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }


// ---------------------------------------------------------------------------------------------------------------
// fine matching (LoFTR FineMatching, reached from LoFTR_matcher.match, etc/feature_matching_baselines/matchers.py:50-55 -> mkpts1_f):
// per matched window m: sim[r] = <g0[m, centre], g1[m, r]> / sqrt(C) over the W*W taps, heat = softmax(sim), sub-pixel offset =
// spatial expectation of heat over the normalised grid linspace(-1, 1, W)^2 (kornia dsnt.spatial_expectation2d), and
// mkpts1_f = mkpts1_c + offset * (W // 2) * (image / fine-map scale).  One wavefront per window: the two feature rows of a tap are
// read once (coalesced, C / 64 values per lane), the W*W dot products are wave reductions, the softmax and the expectation live in
// lanes 0 .. W*W - 1.  The result goes straight into the correspondence tensor at the match's slot.
__global__ void __launch_bounds__(256) fine_match_kernel(const float *__restrict__ g0, const float *__restrict__ g1, int ld, int C, int M, int W,
                                                         float inv_sqrt_c, float out_scale, const int32_t *__restrict__ lin_idx,
                                                         const float *__restrict__ k1, float *__restrict__ pts1, float *__restrict__ expec)
{
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const int WW = W * W;
    const float *pc = g0 + ((size_t)m * WW + WW / 2) * ld;
    float mine = -3.0e38f;
    for (int r = 0; r < WW; ++r) {
        const float *q = g1 + ((size_t)m * WW + r) * ld;
        float sacc = 0.f;
        for (int c = lane; c < C; c += 64) sacc = sacc + pc[c] * q[c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sacc = sacc + __shfl_xor(sacc, off, 64);
        if (lane == r) mine = sacc * inv_sqrt_c;
    }
    float mx = mine;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float e = (lane < WW) ? __expf(mine - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum = sum + __shfl_xor(sum, off, 64);
    const float heat = e / sum;
    const float step = (W > 1) ? 2.0f / (float)(W - 1) : 0.f;
    float cx = (lane < WW) ? heat * (-1.0f + step * (float)(lane % W)) : 0.f;
    float cy = (lane < WW) ? heat * (-1.0f + step * (float)(lane / W)) : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { cx = cx + __shfl_xor(cx, off, 64); cy = cy + __shfl_xor(cy, off, 64); }
    if (lane == 0) {
        if (expec) { expec[2 * (size_t)m] = cx; expec[2 * (size_t)m + 1] = cy; }
        if (pts1) {
            const size_t o = 2 * (size_t)lin_idx[m];
            pts1[o] = k1[o] + cx * out_scale; pts1[o + 1] = k1[o + 1] + cy * out_scale;
        }
    }
}

extern "C" {

size_t mfr_loftr_linear_attention_workspace_bytes(int B, int L, int heads)
{
    if (B <= 0 || L <= 0 || heads <= 0) return 0;
    const size_t nch = (size_t)(L + LA_CHUNK - 1) / LA_CHUNK;
    return align_up((size_t)B * heads * nch * 1024 * sizeof(float), 256) + align_up((size_t)B * heads * nch * 32 * sizeof(float), 256);
}

// q [B,L,ldq], k,v [B,L,ld] (head h = channels [32h, 32h+32)); out [B,L,ldo]
int mfr_loftr_linear_attention(const float *q, int ldq, const float *k, const float *v, int ld, int B, int L, int heads,
                               void *workspace, size_t workspace_bytes, float *out, int ldo, void *stream)
{
    if (!q || !k || !v || !out || !workspace || B <= 0 || L <= 0 || heads <= 0 || (ldq & 3)) return MFR_E_ARG;
    if (workspace_bytes < mfr_loftr_linear_attention_workspace_bytes(B, L, heads)) return MFR_E_WORKSPACE;
    const int nch = (L + LA_CHUNK - 1) / LA_CHUNK;
    float *kvp = (float *)workspace;
    float *ksp = (float *)((char *)workspace + align_up((size_t)B * heads * nch * 1024 * sizeof(float), 256));
    hipStream_t s = (hipStream_t)stream;
    const float vlen = (float)L;
    hipLaunchKernelGGL(la_kv_kernel, dim3(nch, heads, B), dim3(256), 0, s, k, v, ld, L, 1.0f / vlen, kvp, ksp);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(la_fold_kernel, dim3(heads, B), dim3(256), 0, s, nch, kvp, ksp);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(la_out_kernel, dim3((L + 127) / 128, heads, B), dim3(256), 0, s, q, ldq, L, nch, 1, kvp, ksp, vlen, out, ldo);
    CHECK_LAUNCH();
    return 0;
}

size_t mfr_loftr_coarse_match_workspace_bytes(int B, int L0, int L1)
{
    if (B <= 0 || L0 <= 0 || L1 <= 0) return 0;
    const size_t nst = (size_t)(L0 + DSM_RS - 1) / DSM_RS, ncb = (size_t)(L1 + DSM_CB - 1) / DSM_CB;
    return align_up((size_t)B * L0 * 4, 256) * 4 + align_up((size_t)B * L1 * 4, 256) * 3 +
           2 * align_up((size_t)B * ncb * L0 * 4, 256) + 2 * align_up((size_t)B * nst * L1 * 4, 256);
}

// S [B,L0,L1] = (f0/sqrt(C)) (f1/sqrt(C))^T (temperature applied here) -> matches (i ascending)
// variant 0: two sweeps over S (tile kernels + folds); variant 1: the round-1 four sweeps (row / column kernels) -- same
// definition of every quantity, kept for A/B timing and as a cross-check of the tiled reductions
int mfr_loftr_coarse_match_variant(const float *S, int B, int h0, int w0, int h1, int w1, float temperature, float thr, int border,
                                   void *workspace, size_t workspace_bytes, int32_t *i_ids, int32_t *j_ids, float *mconf,
                                   int32_t *n_match, int variant, void *stream)
{
    if (!S || !workspace || !i_ids || !j_ids || !mconf || !n_match || B <= 0 || h0 <= 0 || w0 <= 0 || h1 <= 0 || w1 <= 0)
        return MFR_E_ARG;
    const int L0 = h0 * w0, L1 = h1 * w1;
    if (workspace_bytes < mfr_loftr_coarse_match_workspace_bytes(B, L0, L1)) return MFR_E_WORKSPACE;
    char *ws = (char *)workspace;
    const size_t a0 = align_up((size_t)B * L0 * 4, 256), a1 = align_up((size_t)B * L1 * 4, 256);
    float *rmax = (float *)ws, *rsum = (float *)(ws + a0), *rbest = (float *)(ws + 2 * a0);
    int *rarg = (int *)(ws + 3 * a0);
    float *cmax = (float *)(ws + 4 * a0), *csum = (float *)(ws + 4 * a0 + a1), *cbest = (float *)(ws + 4 * a0 + 2 * a1);
    hipStream_t s = (hipStream_t)stream;
    if (variant != 0 && variant != 1) return MFR_E_ARG;
    if (variant == 1) {
        hipLaunchKernelGGL(dsm_rowstat_kernel, dim3((L0 + 3) / 4, B), dim3(256), 0, s, S, L0, L1, temperature, rmax, rsum);
        hipLaunchKernelGGL(dsm_colstat_kernel, dim3((L1 + 63) / 64, B), dim3(1024), 0, s, S, L0, L1, temperature, cmax, csum);
        CHECK_LAUNCH();
        hipLaunchKernelGGL(dsm_rowbest_kernel, dim3((L0 + 3) / 4, B), dim3(256), 0, s, S, L0, L1, temperature, rmax, rsum, cmax,
                           csum, rbest, rarg);
        hipLaunchKernelGGL(dsm_colbest_kernel, dim3((L1 + 63) / 64, B), dim3(1024), 0, s, S, L0, L1, temperature, rmax, rsum,
                           cmax, csum, cbest);
        CHECK_LAUNCH();
    } else {
        const int nst = (L0 + DSM_RS - 1) / DSM_RS, ncb = (L1 + DSM_CB - 1) / DSM_CB;
        if (B > 65535 || ncb > 65535) return MFR_E_ARG;
        const size_t ar = align_up((size_t)B * ncb * L0 * 4, 256), ac = align_up((size_t)B * nst * L1 * 4, 256);
        char *pw = ws + 4 * a0 + 3 * a1;
        float *rp0 = (float *)pw, *rp1 = (float *)(pw + ar), *cp0 = (float *)(pw + 2 * ar), *cp1 = (float *)(pw + 2 * ar + ac);
        const int nfold = (max(L0, L1) + 255) / 256;
        hipLaunchKernelGGL(dsm_stats_tile_kernel, dim3(nst, ncb, B), dim3(256), 0, s, S, L0, L1, temperature, rp0, rp1, cp0, cp1);
        hipLaunchKernelGGL(dsm_stats_fold_kernel, dim3(nfold, B), dim3(256), 0, s, L0, L1, nst, ncb, rp0, rp1, cp0, cp1, rmax, rsum,
                           cmax, csum);
        CHECK_LAUNCH();
        hipLaunchKernelGGL(dsm_best_tile_kernel, dim3(nst, ncb, B), dim3(256), 0, s, S, L0, L1, temperature, rmax, rsum, cmax, csum,
                           rp0, (int *)rp1, cp0);
        hipLaunchKernelGGL(dsm_best_fold_kernel, dim3(nfold, B), dim3(256), 0, s, L0, L1, nst, ncb, rp0, (const int *)rp1, cp0, rbest,
                           rarg, cbest);
        CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(dsm_match_kernel, dim3(B), dim3(256), 0, s, L0, L1, h0, w0, h1, w1, thr, border, rbest, rarg, cbest,
                       i_ids, j_ids, mconf, n_match);
    CHECK_LAUNCH();
    return 0;
}

int mfr_loftr_coarse_match(const float *S, int B, int h0, int w0, int h1, int w1, float temperature, float thr, int border,
                           void *workspace, size_t workspace_bytes, int32_t *i_ids, int32_t *j_ids, float *mconf,
                           int32_t *n_match, void *stream)
{
    return mfr_loftr_coarse_match_variant(S, B, h0, w0, h1, w1, temperature, thr, border, workspace, workspace_bytes, i_ids, j_ids,
                                          mconf, n_match, 0, stream);
}

// feat [Bimg,Hf,Wf,C] NHWC -> out [M, win*win, C] windows centred on coarse cell `cell_ids[m]` of image `img_ids[m]`
int mfr_loftr_gather_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids,
                             const int32_t *cell_ids, int M, int wc, int stride, int win, float *out, void *stream)
{
    if (!feat || !img_ids || !cell_ids || !out || Bimg <= 0 || Hf <= 0 || Wf <= 0 || C <= 0 || M < 0 || wc <= 0) return MFR_E_ARG;
    if (M == 0) return 0;
    hipLaunchKernelGGL(fine_gather_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, feat, Hf, Wf, C, img_ids, cell_ids, wc,
                       stride, win, out);
    CHECK_LAUNCH();
    return 0;
}

// q [Bw,25,ldq], k,v [Bw,25,ld], out [Bw,25,ldo]; d_model 128 (8 heads x 16), 25 tokens per window
int mfr_loftr_fine_attention(const float *q, int ldq, const float *k, const float *v, int ld, int Bw, int L, int D, int heads,
                             float *out, int ldo, void *stream)
{
    if (!q || !k || !v || !out || Bw < 0 || L != FA_L || D != FA_D || heads != 8 || ldq < D || ld < D || ldo < D) return MFR_E_ARG;
    if (Bw == 0) return 0;
    hipLaunchKernelGGL(fine_attention_kernel, dim3((Bw + 1) / 2), dim3(128), 0, (hipStream_t)stream, q, ldq, k, v, ld, Bw, out, ldo);
    CHECK_LAUNCH();
    return 0;
}

// g0, g1 [M, W*W, ld] (first C columns are the fine features of view 0 / view 1), lin_idx [M] (slot of the match in the flattened
// [B * L0] correspondence layout), k1 / pts1 [B * L0, 2]: pts1[slot] = k1[slot] + expectation * out_scale; expec [M, 2] optional
// (the normalised expectation itself); pts1 / lin_idx / k1 may be NULL when only expec is wanted.
int mfr_loftr_fine_match(const float *g0, const float *g1, int ld, int C, int M, int W, float out_scale, const int32_t *lin_idx,
                         const float *k1, float *pts1, float *expec, void *stream)
{
    if (!g0 || !g1 || ld < C || C <= 0 || M < 0 || W < 1 || W * W > 64 || (!pts1 && !expec) || (pts1 && (!lin_idx || !k1))) return MFR_E_ARG;
    if (M == 0) return 0;
    hipLaunchKernelGGL(fine_match_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, g0, g1, ld, C, M, W, 1.0f / sqrtf((float)C),
                       out_scale, lin_idx, k1, pts1, expec);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
