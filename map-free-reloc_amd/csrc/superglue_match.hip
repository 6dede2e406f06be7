// superglue_match.hip -- SuperGlue optimal-transport matching head on gfx950.
//
// Reference call site: SuperGlue_matcher.match (etc/feature_matching_baselines/matchers.py:93-120)
// -> upstream SuperGlue.forward tail (un-vendored; algorithm per SURVEY.md Appendix A.3):
//   scores = mdesc0^T mdesc1 / sqrt(256)  ->  log_optimal_transport(scores, bin_score, iters=20)
//   ->  row/col argmax over the non-dustbin block, mutual check, exp(score) > match_threshold (0.2)
//   ->  mkpts0 = kpts0[valid], mkpts1 = kpts1[matches[valid]]            (matchers.py:111-116)
//
// The (m+1) x (n+1) coupling matrix with its dustbin row/column is never materialised: the
// dustbin entries all equal alpha, so the row pass adds one analytic term and the dustbin row /
// column are handled in closed form.  Only the raw m x n score block is read.
//
//   sg_row_kernel   u_i = log_mu_i - LSE_j(S_ij + v_j)  (one wavefront per row, coalesced)
//   sg_col_kernel   v_j = log_nu_j - LSE_i(S_ij + u_i)  (64 columns per workgroup, rows split over
//                   16 wavefront-rows, online-LSE merge through LDS)
//   sg_match_kernel row/col argmax of Z = S + u + v - norm, mutual check, threshold, ordered
//                   compaction into the [B, maxN, 2] x2 correspondence layout of the solver C-ABI.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

// u / v row stride: ldS + 1 entries (dustbin) padded to a multiple of 4 floats so that rows of v are 16-byte aligned
#define SG_LDV(ldS) (((ldS) + 4) & ~3)

struct Lse { float m, s; };
static __device__ __forceinline__ void lse_add(Lse &a, float x)
{
    if (x > a.m) { a.s = a.s * __expf(a.m - x) + 1.f; a.m = x; }
    else a.s += __expf(x - a.m);
}
static __device__ __forceinline__ void lse_merge(Lse &a, float m, float s)
{
    if (m == -INFINITY) return;
    if (m > a.m) { a.s = a.s * __expf(a.m - m) + s; a.m = m; }
    else a.s += s * __expf(m - a.m);
}
static __device__ __forceinline__ float lse_val(const Lse &a) { return a.m + __logf(a.s); }

// rows 0..m (row m = dustbin row).  S: [B, ldS, ldS] raw scores (already / sqrt(256)).  One wavefront per row; a lane takes
// float4 pieces (16-byte loads of the row and of v, all issued before the first use), reduces them with max-then-sum (no
// data-dependent branch per element) and the 64 partial (max, sum) pairs merge in a butterfly.
__global__ void __launch_bounds__(256) sg_row_kernel(const float *__restrict__ S, int ldS, const int *__restrict__ n0,
                                                     const int *__restrict__ n1, float alpha,
                                                     const float *__restrict__ v /*[B, SG_LDV]*/,
                                                     float *__restrict__ u /*[B, SG_LDV]*/)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = n0[b], n = n1[b];
    if (i > m || m == 0 || n == 0) return;
    const float *vb = v + (size_t)b * SG_LDV(ldS);
    const float norm = -__logf((float)(m + n));
    Lse a = { -INFINITY, 0.f };
    if (i < m && !(ldS & 3) && ldS <= 1024) {
        const float4 *row4 = (const float4 *)(S + ((size_t)b * ldS + i) * ldS), *v4 = (const float4 *)vb;
        float x[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j4 = lane + 64 * k;                         // float4 index: columns 4 j4 .. 4 j4 + 3
            float4 r = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * j4 < n) { r = row4[j4]; w = v4[j4]; }
            x[4 * k] = (4 * j4 < n) ? r.x + w.x : -INFINITY;
            x[4 * k + 1] = (4 * j4 + 1 < n) ? r.y + w.y : -INFINITY;
            x[4 * k + 2] = (4 * j4 + 2 < n) ? r.z + w.z : -INFINITY;
            x[4 * k + 3] = (4 * j4 + 3 < n) ? r.w + w.w : -INFINITY;
        }
        float mx = x[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) mx = fmaxf(mx, x[k]);
        if (mx > -INFINITY) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += __expf(x[k] - mx);
            a.m = mx; a.s = sum;
        }
    } else if (i < m) {
        const float *row = S + ((size_t)b * ldS + i) * ldS;
        for (int j = lane; j < n; j += 64) lse_add(a, row[j] + vb[j]);
    } else {
        for (int j = lane; j < n; j += 64) lse_add(a, alpha + vb[j]);
    }
    if (lane == 0) lse_add(a, alpha + vb[n]);                     // dustbin column
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float om = __shfl_xor(a.m, off, 64), os = __shfl_xor(a.s, off, 64);
        lse_merge(a, om, os);
    }
    if (lane == 0) {
        const float log_mu = (i < m) ? norm : (__logf((float)n) + norm);
        u[(size_t)b * SG_LDV(ldS) + i] = log_mu - lse_val(a);
    }
}

// columns 0..n (column n = dustbin column); block = 64 columns x 16 row-groups
__global__ void __launch_bounds__(1024) sg_col_kernel(const float *__restrict__ S, int ldS, const int *__restrict__ n0,
                                                      const int *__restrict__ n1, float alpha,
                                                      const float *__restrict__ u, float *__restrict__ v)
{
    __shared__ float sm[16][64], ss[16][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int m = n0[b], n = n1[b];
    if (m == 0 || n == 0) return;
    const float *ub = u + (size_t)b * SG_LDV(ldS);
    Lse a = { -INFINITY, 0.f };
    if (j <= n) {
        if (j < n) {
            // four independent accumulators: four rows in flight per lane instead of one dependent exp chain
            const float *col = S + (size_t)b * ldS * ldS + j;
            Lse a1 = { -INFINITY, 0.f }, a2 = a1, a3 = a1;
            int i = g;
            for (; i + 48 < m; i += 64) {
                const float x0 = col[(size_t)i * ldS] + ub[i], x1 = col[(size_t)(i + 16) * ldS] + ub[i + 16];
                const float x2 = col[(size_t)(i + 32) * ldS] + ub[i + 32], x3 = col[(size_t)(i + 48) * ldS] + ub[i + 48];
                lse_add(a, x0); lse_add(a1, x1); lse_add(a2, x2); lse_add(a3, x3);
            }
            for (; i < m; i += 16) lse_add(a, col[(size_t)i * ldS] + ub[i]);
            lse_merge(a, a1.m, a1.s); lse_merge(a2, a3.m, a3.s); lse_merge(a, a2.m, a2.s);
        } else {
            for (int i = g; i < m; i += 16) lse_add(a, alpha + ub[i]);
        }
        if (g == 0) lse_add(a, alpha + ub[m]);                     // dustbin row
    }
    sm[g][lane] = a.m; ss[g][lane] = a.s;
    __syncthreads();
    if (g == 0 && j <= n) {
        for (int k = 1; k < 16; ++k) lse_merge(a, sm[k][lane], ss[k][lane]);
        const float norm = -__logf((float)(m + n));
        const float log_nu = (j < n) ? norm : (__logf((float)m) + norm);
        v[(size_t)b * SG_LDV(ldS) + j] = log_nu - lse_val(a);
    }
}

// per-row argmax over j<n of (S_ij + v_j) and per-column argmax over i<m of (S_ij + u_i)
__global__ void __launch_bounds__(256) sg_rowmax_kernel(const float *__restrict__ S, int ldS,
                                                        const int *__restrict__ n0, const int *__restrict__ n1,
                                                        const float *__restrict__ v, int *__restrict__ idx0,
                                                        float *__restrict__ val0)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = n0[b], n = n1[b];
    if (i >= m || n == 0) return;
    const float *row = S + ((size_t)b * ldS + i) * ldS;
    const float *vb = v + (size_t)b * SG_LDV(ldS);
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const float x = row[j] + vb[j];
        if (x > best) { best = x; bi = j; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { idx0[(size_t)b * ldS + i] = bi; val0[(size_t)b * ldS + i] = best; }
}

__global__ void __launch_bounds__(1024) sg_colmax_kernel(const float *__restrict__ S, int ldS,
                                                         const int *__restrict__ n0, const int *__restrict__ n1,
                                                         const float *__restrict__ u, int *__restrict__ idx1)
{
    __shared__ float sb[16][64];
    __shared__ int si[16][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int m = n0[b], n = n1[b];
    if (m == 0 || n == 0) return;
    const float *ub = u + (size_t)b * SG_LDV(ldS);
    float best = -INFINITY; int bi = 0x7fffffff;
    if (j < n) {
        const float *col = S + (size_t)b * ldS * ldS + j;
        for (int i = g; i < m; i += 16) {
            const float x = col[(size_t)i * ldS] + ub[i];
            if (x > best) { best = x; bi = i; }
        }
    }
    sb[g][lane] = best; si[g][lane] = bi;
    __syncthreads();
    if (g == 0 && j < n) {
        for (int k = 1; k < 16; ++k) {
            const float ob = sb[k][lane]; const int oi = si[k][lane];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        idx1[(size_t)b * ldS + j] = bi;
    }
}

// one workgroup per pair: mutual check + threshold + ordered compaction (matchers.py:111-116)
__global__ void __launch_bounds__(256) sg_match_kernel(
    int ldS, const int *__restrict__ n0, const int *__restrict__ n1, const float *__restrict__ u,
    const int *__restrict__ idx0, const float *__restrict__ val0, const int *__restrict__ idx1, float thr,
    const float *__restrict__ kpts0, const float *__restrict__ kpts1, int K /*kpts stride*/,
    int *__restrict__ matches0, float *__restrict__ mscores0, float *__restrict__ pts0, float *__restrict__ pts1,
    int maxN, int *__restrict__ n_corr)
{
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = n0[b], n = n1[b];
    if (tid == 0) base_s = 0;
    __syncthreads();
    const float norm = (m > 0 && n > 0) ? -__logf((float)(m + n)) : 0.f;
    for (int start = 0; start < ldS; start += 256) {
        const int i = start + tid;
        bool valid = false;
        int j = -1;
        float sc = 0.f;
        if (i < m && n > 0) {
            j = idx0[(size_t)b * ldS + i];
            const bool mutual = idx1[(size_t)b * ldS + j] == i;
            // Z_ij = S_ij + u_i + v_j - norm ; val0 = S_ij + v_j
            const float z = val0[(size_t)b * ldS + i] + u[(size_t)b * SG_LDV(ldS) + i] - norm;
            sc = mutual ? __expf(z) : 0.f;
            valid = mutual && (sc > thr);
        }
        if (i < ldS) {
            matches0[(size_t)b * ldS + i] = valid ? j : -1;
            mscores0[(size_t)b * ldS + i] = (i < m) ? sc : 0.f;
        }
        const unsigned long long bal = __ballot(valid);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wid; ++w) off += wave_cnt[w];
        if (valid) {
            const int o = off + wpre;
            if (o < maxN) {
                pts0[((size_t)b * maxN + o) * 2] = kpts0[((size_t)b * K + i) * 2];
                pts0[((size_t)b * maxN + o) * 2 + 1] = kpts0[((size_t)b * K + i) * 2 + 1];
                pts1[((size_t)b * maxN + o) * 2] = kpts1[((size_t)b * K + j) * 2];
                pts1[((size_t)b * maxN + o) * 2 + 1] = kpts1[((size_t)b * K + j) * 2 + 1];
            }
        }
        __syncthreads();
        if (tid == 0) base_s = off + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_corr[b] = min(base_s, maxN);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
struct SgWs { size_t u, v, idx0, val0, idx1, total; };
static SgWs sg_ws_layout(int B, int ldS)
{
    SgWs w; size_t o = 0;
    w.u = o;    o = align_up(o + sizeof(float) * (size_t)B * SG_LDV(ldS), 256);
    w.v = o;    o = align_up(o + sizeof(float) * (size_t)B * SG_LDV(ldS), 256);
    w.idx0 = o; o = align_up(o + sizeof(int) * (size_t)B * ldS, 256);
    w.val0 = o; o = align_up(o + sizeof(float) * (size_t)B * ldS, 256);
    w.idx1 = o; o = align_up(o + sizeof(int) * (size_t)B * ldS, 256);
    w.total = o;
    return w;
}

extern "C" {

size_t mfr_sg_match_workspace_bytes(int B, int ldS)
{
    if (B <= 0 || ldS <= 0) return 0;
    return sg_ws_layout(B, ldS).total;
}

int mfr_sg_sinkhorn_match(const float *S, int B, int ldS, const int32_t *n0, const int32_t *n1,
                          float bin_score, int iters, float match_thr,
                          const float *kpts0, const float *kpts1, int K,
                          void *workspace, size_t workspace_bytes,
                          int32_t *matches0, float *mscores0, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                          void *stream)
{
    if (!S || !n0 || !n1 || !kpts0 || !kpts1 || !workspace || !matches0 || !mscores0 || !pts0 || !pts1 || !n_corr ||
        B <= 0 || ldS <= 0 || K < ldS || maxN <= 0 || iters < 0) return MFR_E_ARG;
    const SgWs w = sg_ws_layout(B, ldS);
    if (workspace_bytes < w.total) return MFR_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *u = (float *)(ws + w.u), *v = (float *)(ws + w.v), *val0 = (float *)(ws + w.val0);
    int *idx0 = (int *)(ws + w.idx0), *idx1 = (int *)(ws + w.idx1);
    // u = v = 0 (upstream log_sinkhorn_iterations)
    if (mfr_zero_async(ws + w.u, w.idx0 - w.u, s) != hipSuccess) return MFR_E_LAUNCH;
    const dim3 rgrid((ldS + 1 + 3) / 4, B), cgrid((ldS + 1 + 63) / 64, B);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(sg_row_kernel, rgrid, dim3(256), 0, s, S, ldS, n0, n1, bin_score, v, u);
        hipLaunchKernelGGL(sg_col_kernel, cgrid, dim3(1024), 0, s, S, ldS, n0, n1, bin_score, u, v);
    }
    CHECK_LAUNCH();
    hipLaunchKernelGGL(sg_rowmax_kernel, dim3((ldS + 3) / 4, B), dim3(256), 0, s, S, ldS, n0, n1, v, idx0, val0);
    hipLaunchKernelGGL(sg_colmax_kernel, dim3((ldS + 63) / 64, B), dim3(1024), 0, s, S, ldS, n0, n1, u, idx1);
    hipLaunchKernelGGL(sg_match_kernel, dim3(B), dim3(256), 0, s, ldS, n0, n1, u, idx0, val0, idx1, match_thr, kpts0,
                       kpts1, K, matches0, mscores0, pts0, pts1, maxN, n_corr);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
