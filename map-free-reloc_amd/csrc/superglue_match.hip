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
// online log-sum-exp, BRANCH-FREE: with nm = max(a.m, x) the update a.s * exp(a.m - nm) + exp(x - nm) is the textbook two-case rule in one
// expression -- whichever exponent is zero gives exactly 1 (v_exp_f32(0) = 1, a product with 1 and an fma with 1 are exact), so no bit differs
// from `if (x > a.m) a.s = a.s * exp(a.m - x) + 1 else a.s += exp(x - a.m)`, but a wavefront whose lanes disagree on the case no longer runs
// both sides under exec masks (rounds 1-3: ~25 instructions per element in sg_col_kernel, which made that pass instruction-bound).
// x is finite at every call site; a.m = -inf (empty) gives exp(-inf) = 0 times a.s = 0.
static __device__ __forceinline__ void lse_add(Lse &a, float x)
{
    const float nm = fmaxf(a.m, x);
    a.s = __builtin_fmaf(a.s, __expf(a.m - nm), __expf(x - nm));
    a.m = nm;
}
// merge of a partial (m, s); an empty partial (m = -inf) leaves a unchanged (select, not a branch: the discarded side may hold a NaN)
static __device__ __forceinline__ void lse_merge(Lse &a, float m, float s)
{
    const float nm = fmaxf(a.m, m);
    const float t = s * __expf(m - nm);
    const float ns = __builtin_fmaf(a.s, __expf(a.m - nm), t);
    const bool keep = m == -INFINITY;
    a.s = keep ? a.s : ns;
    a.m = keep ? a.m : nm;
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
        // four independent accumulators: four rows in flight per lane instead of one dependent exp chain.  The dustbin column (j = n, every
        // entry alpha) takes the same path (round 4: it was one chain), so that sg_sweep_kernel can treat it as a 17th column of a lane.
        const bool dust = j == n;
        const float *col = S + (size_t)b * ldS * ldS + (dust ? 0 : j);
        Lse a1 = { -INFINITY, 0.f }, a2 = a1, a3 = a1;
        int i = g;
        for (; i + 48 < m; i += 64) {
            const float s0 = dust ? alpha : col[(size_t)i * ldS], s1 = dust ? alpha : col[(size_t)(i + 16) * ldS];
            const float s2 = dust ? alpha : col[(size_t)(i + 32) * ldS], s3 = dust ? alpha : col[(size_t)(i + 48) * ldS];
            lse_add(a, s0 + ub[i]); lse_add(a1, s1 + ub[i + 16]); lse_add(a2, s2 + ub[i + 32]); lse_add(a3, s3 + ub[i + 48]);
        }
        for (; i < m; i += 16) lse_add(a, (dust ? alpha : col[(size_t)i * ldS]) + ub[i]);
        lse_merge(a, a1.m, a1.s); lse_merge(a2, a3.m, a3.s); lse_merge(a, a2.m, a2.s);
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

// ---- round 4: ONE sweep over S per Sinkhorn iteration -------------------------------------------------------------------------------------
// sg_row_kernel + sg_col_kernel stream the (MALL-resident) 4 MB score block of every pair twice per iteration; at 32 pairs both passes run at
// the rate the 134 MB come out of the cache (25.6 + 31.2 us), not at the exp rate.  sg_sweep_kernel reads a row ONCE: a wavefront computes
// u_i from the row exactly as sg_row_kernel does (same lane -> column assignment, same order of sums, lane 0's merge result) and then, with
// the row still in its registers, adds S_ij + u_i to the column partials.  The partials are kept in sg_col_kernel's own order, so that v --
// and with it every match index and score -- is the same bits as the two-pass version: that kernel gives row i to row group g = i mod 16 and,
// inside a group, to accumulator (i / 16) mod 4 while whole quads of rows remain, the left-over rows to accumulator 0.  Here workgroup g of a
// pair is row group g and its four wavefronts are the four accumulators (each lane holds the partials of its 16 columns); they merge as there
// (a += a1, a2 += a3, a += a2), workgroup 0 adds the dustbin row, and the 16 group partials go to a small buffer that sg_colmerge_kernel folds
// in sg_col_kernel's order (k = 1 .. 15) into v.  The dustbin column is a 17th column of every lane (its entries are all alpha), and the
// dustbin row's u_m is computed by workgroup 0 before its rows.  Requires the 16-byte fast path of sg_row_kernel (ldS % 4 == 0, ldS <= 1024); other shapes keep
// the two-pass kernels.
__global__ void __launch_bounds__(256) sg_sweep_kernel(const float *__restrict__ S, int ldS, const int *__restrict__ n0, const int *__restrict__ n1,
                                                       float alpha, const float *__restrict__ v, float *__restrict__ u,
                                                       float *__restrict__ part_m /*[B, 16, SG_LDV]*/, float *__restrict__ part_s)
{
    __shared__ float xm[2][17][64], xs[2][17][64];
    __shared__ float um_s;
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int m = n0[b], n = n1[b];
    if (m == 0 || n == 0) return;
    const int ldv = SG_LDV(ldS);
    const float *vb = v + (size_t)b * ldv;
    float *ub = u + (size_t)b * ldv;
    const float norm = -__logf((float)(m + n));
    const float4 *v4 = (const float4 *)vb;
    const float vn = alpha + vb[n];                                   // dustbin column term of every row

    // the dustbin row's u (row m: every entry alpha), by wavefront 0 of workgroup 0 -- sg_row_kernel's third branch
    if (g == 0) {
        if (r == 0) {
            Lse a = { -INFINITY, 0.f };
            for (int j = lane; j < n; j += 64) lse_add(a, alpha + vb[j]);
            if (lane == 0) lse_add(a, vn);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float om = __shfl_xor(a.m, off, 64), os = __shfl_xor(a.s, off, 64);
                lse_merge(a, om, os);
            }
            if (lane == 0) { const float um = (__logf((float)n) + norm) - lse_val(a); ub[m] = um; um_s = um; }
        }
        __syncthreads();
    }

    Lse acc[17];                                                      // [16]: the dustbin column (every lane the same value)
#pragma unroll
    for (int k = 0; k < 17; ++k) { acc[k].m = -INFINITY; acc[k].s = 0.f; }
    // whole quads of rows of this group: rows g + 64 q + 16 r while g + 64 q + 48 < m; then the left-over rows (step 16) on wavefront 0
    const int nq = (m - 48 - g > 0) ? (m - 48 - g + 63) / 64 : 0;
    const int ntail = (r == 0) ? max(0, (m - (g + 64 * nq) + 15) / 16) : 0;
    const int nrows = nq + ntail;
    auto row_of = [&](int t) { return t < nq ? g + 64 * t + 16 * r : g + 64 * nq + 16 * (t - nq); };
    // four rows in flight per wavefront (a row is 4 KB; one row ahead left the sweep bound by the cache latency): ring of four register sets
    float4 rb0[4], rb1[4], rb2[4], rb3[4];
    // No per-element branches: every lane loads its four 16-byte pieces (index clamped into the row), columns >= n are masked to -inf by
    // selects for the row sum and simply accumulate into column partials that nobody reads (a column's partial never mixes with another's).
    const int j4max = (ldS >> 2) - 1;
    bool ok[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ok[k] = 4 * (lane + 64 * (k >> 2)) + (k & 3) < n;
    auto load_row = [&](float4 (&d)[4], int t) {
        if (t >= nrows) return;
        const float4 *row4 = (const float4 *)(S + ((size_t)b * ldS + row_of(t)) * ldS);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = row4[min(lane + 64 * k, j4max)];
    };
    float4 w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = v4[min(lane + 64 * k, j4max)];
    auto do_row = [&](const float4 (&cur)[4], int t) {
        if (t >= nrows) return;
        const int i = row_of(t);
        // ---- u_i: sg_row_kernel's fast path (same sums in the same order)
        float x[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[4 * k] = ok[4 * k] ? cur[k].x + w[k].x : -INFINITY;
            x[4 * k + 1] = ok[4 * k + 1] ? cur[k].y + w[k].y : -INFINITY;
            x[4 * k + 2] = ok[4 * k + 2] ? cur[k].z + w[k].z : -INFINITY;
            x[4 * k + 3] = ok[4 * k + 3] ? cur[k].w + w[k].w : -INFINITY;
        }
        float mx = x[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) mx = fmaxf(mx, x[k]);
        const float mxs = (mx > -INFINITY) ? mx : 0.f;              // a lane without valid columns: every term exp(-inf - 0) = 0, (m, s) = (-inf, 0)
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += __expf(x[k] - mxs);
        Lse a = { mx, sum };
        {
            Lse a0 = a;
            lse_add(a0, vn);                                        // dustbin column: lane 0 only
            a.m = lane == 0 ? a0.m : a.m; a.s = lane == 0 ? a0.s : a.s;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float om = __shfl_xor(a.m, off, 64), os = __shfl_xor(a.s, off, 64);
            lse_merge(a, om, os);
        }
        const float ui = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, norm - lse_val(a))));   // lane 0's, as stored
        if (lane == 0) ub[i] = ui;
        // ---- column partials of this lane's 16 columns: sg_col_kernel's lse_add(S_ij + u_i), rows in ascending order per accumulator
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lse_add(acc[4 * k], cur[k].x + ui); lse_add(acc[4 * k + 1], cur[k].y + ui);
            lse_add(acc[4 * k + 2], cur[k].z + ui); lse_add(acc[4 * k + 3], cur[k].w + ui);
        }
        lse_add(acc[16], alpha + ui);
    };
    load_row(rb0, 0); load_row(rb1, 1); load_row(rb2, 2); load_row(rb3, 3);
    for (int t = 0; t < nrows; t += 4) {
        do_row(rb0, t);     load_row(rb0, t + 4);
        do_row(rb1, t + 1); load_row(rb1, t + 5);
        do_row(rb2, t + 2); load_row(rb2, t + 6);
        do_row(rb3, t + 3); load_row(rb3, t + 7);
    }
    // ---- a += a1, a2 += a3 (wavefronts 1 and 3 hand over), then a += a2
    if (r & 1) {
#pragma unroll
        for (int k = 0; k < 17; ++k) { xm[r >> 1][k][lane] = acc[k].m; xs[r >> 1][k][lane] = acc[k].s; }
    }
    __syncthreads();
    if (!(r & 1)) {
#pragma unroll
        for (int k = 0; k < 17; ++k) lse_merge(acc[k], xm[r >> 1][k][lane], xs[r >> 1][k][lane]);
    }
    __syncthreads();
    if (r == 2) {
#pragma unroll
        for (int k = 0; k < 17; ++k) { xm[0][k][lane] = acc[k].m; xs[0][k][lane] = acc[k].s; }
    }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int k = 0; k < 17; ++k) lse_merge(acc[k], xm[0][k][lane], xs[0][k][lane]);
        if (g == 0) {
            const float xd = alpha + um_s;                            // dustbin row
#pragma unroll
            for (int k = 0; k < 17; ++k) lse_add(acc[k], xd);
        }
        float *pmr = part_m + ((size_t)b * 16 + g) * ldv, *psr = part_s + ((size_t)b * 16 + g) * ldv;
        float4 *pm = (float4 *)pmr, *ps = (float4 *)psr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j4 = lane + 64 * k;
            if (4 * j4 < n) {
                pm[j4] = make_float4(acc[4 * k].m, acc[4 * k + 1].m, acc[4 * k + 2].m, acc[4 * k + 3].m);
                ps[j4] = make_float4(acc[4 * k].s, acc[4 * k + 1].s, acc[4 * k + 2].s, acc[4 * k + 3].s);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);                               // (the float4 holding column n, if any, is written before the scalar overwrite)
        if (lane == (n >> 2) % 64) { pmr[n] = acc[16].m; psr[n] = acc[16].s; }
    }
}

// v_j = log_nu_j - LSE over the 16 group partials (k = 1 .. 15 folded into group 0's, sg_col_kernel's order), columns 0 .. n (n = dustbin)
__global__ void __launch_bounds__(64) sg_colmerge_kernel(int ldS, const int *__restrict__ n0, const int *__restrict__ n1,
                                                         const float *__restrict__ part_m, const float *__restrict__ part_s, float *__restrict__ v)
{
    const int b = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    const int m = n0[b], n = n1[b];
    if (m == 0 || n == 0 || j > n) return;
    const int ldv = SG_LDV(ldS);
    const float norm = -__logf((float)(m + n));
    const float *pm = part_m + (size_t)b * 16 * ldv + j, *ps = part_s + (size_t)b * 16 * ldv + j;
    float mm[16], sv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { mm[k] = pm[(size_t)k * ldv]; sv[k] = ps[(size_t)k * ldv]; }      // all loads before the dependent chain
    Lse a = { mm[0], sv[0] };
#pragma unroll
    for (int k = 1; k < 16; ++k) lse_merge(a, mm[k], sv[k]);
    const float log_nu = (j < n) ? norm : (__logf((float)m) + norm);
    v[(size_t)b * ldv + j] = log_nu - lse_val(a);
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
struct SgWs { size_t u, v, idx0, val0, idx1, part_m, part_s, total; };
static SgWs sg_ws_layout(int B, int ldS)
{
    SgWs w; size_t o = 0;
    w.u = o;    o = align_up(o + sizeof(float) * (size_t)B * SG_LDV(ldS), 256);
    w.v = o;    o = align_up(o + sizeof(float) * (size_t)B * SG_LDV(ldS), 256);
    w.idx0 = o; o = align_up(o + sizeof(int) * (size_t)B * ldS, 256);
    w.val0 = o; o = align_up(o + sizeof(float) * (size_t)B * ldS, 256);
    w.idx1 = o; o = align_up(o + sizeof(int) * (size_t)B * ldS, 256);
    w.part_m = o; o = align_up(o + sizeof(float) * (size_t)B * 16 * SG_LDV(ldS), 256);      // one-sweep iteration: 16 row-group partials per column
    w.part_s = o; o = align_up(o + sizeof(float) * (size_t)B * 16 * SG_LDV(ldS), 256);
    w.total = o;
    return w;
}

extern "C" {

size_t mfr_sg_match_workspace_bytes(int B, int ldS)
{
    if (B <= 0 || ldS <= 0) return 0;
    return sg_ws_layout(B, ldS).total;
}

// variant: 0 = one sweep over S per iteration (sg_sweep_kernel + sg_colmerge_kernel) when ldS % 4 == 0 and ldS <= 1024, else as 1;
// 1 = the two-pass kernels (sg_row_kernel + sg_col_kernel).  Both produce the same bits.
int mfr_sg_sinkhorn_match_variant(const float *S, int B, int ldS, const int32_t *n0, const int32_t *n1,
                                  float bin_score, int iters, float match_thr,
                                  const float *kpts0, const float *kpts1, int K,
                                  void *workspace, size_t workspace_bytes,
                                  int32_t *matches0, float *mscores0, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                                  int variant, void *stream)
{
    if (!S || !n0 || !n1 || !kpts0 || !kpts1 || !workspace || !matches0 || !mscores0 || !pts0 || !pts1 || !n_corr ||
        B <= 0 || ldS <= 0 || K < ldS || maxN <= 0 || iters < 0 || variant < 0 || variant > 1) return MFR_E_ARG;
    const SgWs w = sg_ws_layout(B, ldS);
    if (workspace_bytes < w.total) return MFR_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *u = (float *)(ws + w.u), *v = (float *)(ws + w.v), *val0 = (float *)(ws + w.val0);
    float *part_m = (float *)(ws + w.part_m), *part_s = (float *)(ws + w.part_s);
    int *idx0 = (int *)(ws + w.idx0), *idx1 = (int *)(ws + w.idx1);
    // u = v = 0 (upstream log_sinkhorn_iterations)
    if (mfr_zero_async(ws + w.u, w.idx0 - w.u, s) != hipSuccess) return MFR_E_LAUNCH;
    const bool sweep = variant == 0 && !(ldS & 3) && ldS <= 1024 && !((uintptr_t)S & 15);
    const dim3 rgrid((ldS + 1 + 3) / 4, B), cgrid((ldS + 1 + 63) / 64, B);
    for (int it = 0; it < iters; ++it) {
        if (sweep) {
            hipLaunchKernelGGL(sg_sweep_kernel, dim3(16, B), dim3(256), 0, s, S, ldS, n0, n1, bin_score, v, u, part_m, part_s);
            hipLaunchKernelGGL(sg_colmerge_kernel, cgrid, dim3(64), 0, s, ldS, n0, n1, part_m, part_s, v);
        } else {
            hipLaunchKernelGGL(sg_row_kernel, rgrid, dim3(256), 0, s, S, ldS, n0, n1, bin_score, v, u);
            hipLaunchKernelGGL(sg_col_kernel, cgrid, dim3(1024), 0, s, S, ldS, n0, n1, bin_score, u, v);
        }
    }
    CHECK_LAUNCH();
    hipLaunchKernelGGL(sg_rowmax_kernel, dim3((ldS + 3) / 4, B), dim3(256), 0, s, S, ldS, n0, n1, v, idx0, val0);
    hipLaunchKernelGGL(sg_colmax_kernel, dim3((ldS + 63) / 64, B), dim3(1024), 0, s, S, ldS, n0, n1, u, idx1);
    hipLaunchKernelGGL(sg_match_kernel, dim3(B), dim3(256), 0, s, ldS, n0, n1, u, idx0, val0, idx1, match_thr, kpts0,
                       kpts1, K, matches0, mscores0, pts0, pts1, maxN, n_corr);
    CHECK_LAUNCH();
    return 0;
}

int mfr_sg_sinkhorn_match(const float *S, int B, int ldS, const int32_t *n0, const int32_t *n1,
                          float bin_score, int iters, float match_thr,
                          const float *kpts0, const float *kpts1, int K,
                          void *workspace, size_t workspace_bytes,
                          int32_t *matches0, float *mscores0, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                          void *stream)
{
    return mfr_sg_sinkhorn_match_variant(S, B, ldS, n0, n1, bin_score, iters, match_thr, kpts0, kpts1, K, workspace, workspace_bytes, matches0, mscores0,
                                         pts0, pts1, maxN, n_corr, 0, stream);
}

}  // extern "C"
