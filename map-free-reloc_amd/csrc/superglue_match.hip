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
// round 5: four terms at once -- the block's maximum first, then ONE rescale of the running sum per four terms instead of one per term
// (1.5 exponentials per term instead of 2).  x0 is finite; x1 .. x3 may be -inf (absent: exp(-inf) = 0).
static __device__ __forceinline__ void lse_block4(float x0, float x1, float x2, float x3, float &bm, float &bs)
{
    bm = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
    bs = ((__expf(x0 - bm) + __expf(x1 - bm)) + __expf(x2 - bm)) + __expf(x3 - bm);
}
static __device__ __forceinline__ void lse_merge_block(Lse &a, float bm, float bs)        // bm finite
{
    const float nm = fmaxf(a.m, bm);
    a.s = __builtin_fmaf(a.s, __expf(a.m - nm), bs * __expf(bm - nm));
    a.m = nm;
}
static __device__ __forceinline__ void lse_add4(Lse &a, float x0, float x1, float x2, float x3)
{
    float bm, bs;
    lse_block4(x0, x1, x2, x3, bm, bs);
    lse_merge_block(a, bm, bs);
}
// wavefront-wide max / sum that leave the SAME bits in all 64 lanes (every step pairs two groups and a + b == b + a): DPP inside a row of
// 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then gfx950's v_permlane16_swap / v_permlane32_swap across rows and halves
// (two copies of x go in; one comes back holding the even rows' / lower half's values everywhere, the other the odd rows' / upper half's) --
// no LDS round trip.  Inline asm: the builtin, given the same value twice, was compiled to x + x (hipcc 7.2).
static __device__ __forceinline__ void sg_swap16(float &a, float &b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
static __device__ __forceinline__ void sg_swap32(float &a, float &b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
template <int CTRL> static __device__ __forceinline__ float sg_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
static __device__ __forceinline__ float sg_wave_max(float x)
{
    x = fmaxf(x, sg_dpp<0xB1>(x)); x = fmaxf(x, sg_dpp<0x4E>(x)); x = fmaxf(x, sg_dpp<0x141>(x)); x = fmaxf(x, sg_dpp<0x140>(x));
    float y = x; sg_swap16(x, y); x = fmaxf(x, y);
    y = x; sg_swap32(x, y); x = fmaxf(x, y);
    return x;
}
static __device__ __forceinline__ float sg_wave_sum(float x)
{
    x += sg_dpp<0xB1>(x); x += sg_dpp<0x4E>(x); x += sg_dpp<0x141>(x); x += sg_dpp<0x140>(x);
    float y = x; sg_swap16(x, y); x += y;
    y = x; sg_swap32(x, y); x += y;
    return x;
}
// u_i of a row on the 16-byte fast path: x[k] = S_ij + v_j of this lane's 16 columns (-inf beyond n), vn = alpha + v_n (the dustbin column, a
// 1025th term that lane 0 adds).  Round 5: the row's maximum first (wavefront-wide), then one exponential per term against it and a plain
// wavefront sum -- rounds 1-4 merged 64 (max, sum) pairs in a butterfly of six rescaling steps, 84 instructions a row.
static __device__ __forceinline__ float sg_row_u(const float (&x)[16], float vn, float log_mu, int lane)
{
    float mx = x[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, x[k]);
    mx = fmaxf(sg_wave_max(mx), vn);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += __expf(x[k] - mx);
    const float sd = sum + __expf(vn - mx);
    sum = sg_wave_sum(lane == 0 ? sd : sum);
    return log_mu - (mx + __logf(sum));
}
// rows of accumulator r (0 .. 3) of row group g (= i mod 16), in the order they are added: whole quads first (row g + 64 t + 16 r while
// g + 64 t + 48 < m), then -- accumulator 0 only -- the group's left-over rows (step 16); blocks of four consecutive entries of this list go
// through lse_add4.  Shared by sg_col_kernel and sg_sweep_kernel: the same sums in the same order.
static __device__ __forceinline__ int sg_nq(int m, int g) { return (m - 48 - g > 0) ? (m - 48 - g + 63) / 64 : 0; }
static __device__ __forceinline__ int sg_ntail(int m, int g, int nq) { return max(0, (m - (g + 64 * nq) + 15) / 16); }
static __device__ __forceinline__ int sg_row_of(int g, int r, int nq, int t) { return t < nq ? g + 64 * t + 16 * r : g + 64 * nq + 16 * (t - nq); }

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
        const float ui = sg_row_u(x, alpha + vb[n], norm, lane);
        if (lane == 0) u[(size_t)b * SG_LDV(ldS) + i] = ui;
        return;
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
        const int nq = sg_nq(m, g), ntail = sg_ntail(m, g, nq);
        auto run = [&](Lse &ar, int r) {
            const int nr = nq + (r == 0 ? ntail : 0);
            for (int t = 0; t < nr; t += 4) {
                float x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = sg_row_of(g, r, nq, min(t + q, nr - 1));
                    const float xv = (dust ? alpha : col[(size_t)i * ldS]) + ub[i];
                    x[q] = (t + q < nr) ? xv : -INFINITY;
                }
                lse_add4(ar, x[0], x[1], x[2], x[3]);
            }
        };
        run(a, 0); run(a1, 1); run(a2, 2); run(a3, 3);
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
__global__ void __launch_bounds__(512, 2) sg_sweep_kernel(const float *__restrict__ S, int ldS, const int *__restrict__ n0, const int *__restrict__ n1,
                                                       float alpha, const float *__restrict__ v, float *__restrict__ u,
                                                       float *__restrict__ part_m /*[B, 16, SG_LDV]*/, float *__restrict__ part_s)
{
    __shared__ float xm[1][17][64], xs[1][17][64];
    // the column partials of the four wavefronts (17 per lane: 16 columns + the dustbin column) live in LDS between blocks of rows: the two row
    // sets (128 registers) and the partials (34) do not fit 256 registers together, and one read + one write per FOUR rows is cheap
    __shared__ float4 am4[4][4][64], as4[4][4][64];
    __shared__ float amd[4][64], asd[4][64];
    __shared__ float um_s;
    __shared__ float4 wsh[256];
    // round 5: EIGHT wavefronts -- accumulator r's blocks of four rows alternate between wavefronts (r, 0) and (r, 1); a block's sums (maximum,
    // sum of exponentials: lse_block4) are formed by whichever owns it, the merges into the accumulator happen in block order (lse_merge_block:
    // the even block's wavefront first, a barrier, the odd block's) -- exactly lse_add4's arithmetic, so nothing downstream changes a bit -- and a
    // SIMD holds four wavefronts of ~110 registers instead of two of 204: their dependent exponential chains and row loads fill each other's gaps
    // (profiles/r05_pmc_sinkhorn.json: the four-wavefront kernel spent 34 % of its wave cycles blocked at issue, 28 % parked on loads)
    const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x & 63, w8 = threadIdx.x >> 6, r = w8 & 3, hb = w8 >> 2;
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
        if (w8 == 0) {
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

    if (hb == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { am4[r][k][lane] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); as4[r][k][lane] = make_float4(0.f, 0.f, 0.f, 0.f); }
        amd[r][lane] = -INFINITY; asd[r][lane] = 0.f;                 // the dustbin column (every lane the same value)
    }
    // this wavefront = accumulator r of row group g: its rows t = 0 .. nrows - 1 (sg_row_of), four at a time
    const int nq = sg_nq(m, g);
    const int nrows = nq + (r == 0 ? sg_ntail(m, g, nq) : 0);
    // No per-element branches: every lane loads its four 16-byte pieces (index clamped into the row), columns >= n are masked to -inf by
    // selects for the row sum and simply accumulate into column partials that nobody reads (a column's partial never mixes with another's).
    const int j4max = (ldS >> 2) - 1;
    bool ok[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ok[k] = 4 * (lane + 64 * (k >> 2)) + (k & 3) < n;
    // v of the pair's columns: one LDS copy for the workgroup (16 registers per wavefront otherwise, and the budget is 128)
    if (threadIdx.x < 256) wsh[threadIdx.x] = v4[min((int)threadIdx.x, j4max)];
    // the loop count is the workgroup's (barriers inside): accumulator 0 has the most rows (it takes the group's left-over rows)
    const int nblk0 = (nq + sg_ntail(m, g, nq) + 3) >> 2;
    __syncthreads();                                                  // accumulators initialised
    for (int it = 0; 2 * it < nblk0; ++it) {
        const int t = 4 * (2 * it + hb);
        const bool have = t < nrows;                                  // (wave-uniform)
        float bm[17], bs[17];
        if (have) {
            float4 cur[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (t + q < nrows) {
                    const float4 *row4 = (const float4 *)(S + ((size_t)b * ldS + sg_row_of(g, r, nq, t + q)) * ldS);
#pragma unroll
                    for (int k = 0; k < 4; ++k) cur[q][k] = row4[min(lane + 64 * k, j4max)];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) cur[q][k] = make_float4(0.f, 0.f, 0.f, 0.f);     // absent row: finite filler, its u is -inf below
                }
            }
            float ui[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ui[q] = -INFINITY;
                if (t + q < nrows) {
                    // ---- u_i: sg_row_kernel's fast path (same sums in the same order)
                    float x[16];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 wk = wsh[lane + 64 * k];
                        x[4 * k] = ok[4 * k] ? cur[q][k].x + wk.x : -INFINITY;
                        x[4 * k + 1] = ok[4 * k + 1] ? cur[q][k].y + wk.y : -INFINITY;
                        x[4 * k + 2] = ok[4 * k + 2] ? cur[q][k].z + wk.z : -INFINITY;
                        x[4 * k + 3] = ok[4 * k + 3] ? cur[q][k].w + wk.w : -INFINITY;
                    }
                    ui[q] = sg_row_u(x, vn, norm, lane);
                    if (lane == 0) ub[sg_row_of(g, r, nq, t + q)] = ui[q];
                }
                __builtin_amdgcn_sched_barrier(0);                    // one row's 16 terms at a time: the budget is 128 registers
            }
            // ---- this block's sums of this lane's 16 columns + the dustbin column (sg_col_kernel's lse_add4, first half)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lse_block4(cur[0][k].x + ui[0], cur[1][k].x + ui[1], cur[2][k].x + ui[2], cur[3][k].x + ui[3], bm[4 * k], bs[4 * k]);
                lse_block4(cur[0][k].y + ui[0], cur[1][k].y + ui[1], cur[2][k].y + ui[2], cur[3][k].y + ui[3], bm[4 * k + 1], bs[4 * k + 1]);
                lse_block4(cur[0][k].z + ui[0], cur[1][k].z + ui[1], cur[2][k].z + ui[2], cur[3][k].z + ui[3], bm[4 * k + 2], bs[4 * k + 2]);
                lse_block4(cur[0][k].w + ui[0], cur[1][k].w + ui[1], cur[2][k].w + ui[2], cur[3][k].w + ui[3], bm[4 * k + 3], bs[4 * k + 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            lse_block4(alpha + ui[0], alpha + ui[1], alpha + ui[2], alpha + ui[3], bm[16], bs[16]);
        }
        // ---- merges in block order (second half of lse_add4): the even block, a barrier, the odd block
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (have && hb == ph) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 pm = am4[r][k][lane], ps = as4[r][k][lane];
                    Lse a0 = { pm.x, ps.x }, a1 = { pm.y, ps.y }, a2 = { pm.z, ps.z }, a3 = { pm.w, ps.w };
                    lse_merge_block(a0, bm[4 * k], bs[4 * k]); lse_merge_block(a1, bm[4 * k + 1], bs[4 * k + 1]);
                    lse_merge_block(a2, bm[4 * k + 2], bs[4 * k + 2]); lse_merge_block(a3, bm[4 * k + 3], bs[4 * k + 3]);
                    am4[r][k][lane] = make_float4(a0.m, a1.m, a2.m, a3.m); as4[r][k][lane] = make_float4(a0.s, a1.s, a2.s, a3.s);
                }
                Lse ad = { amd[r][lane], asd[r][lane] };
                lse_merge_block(ad, bm[16], bs[16]);
                amd[r][lane] = ad.m; asd[r][lane] = ad.s;
            }
            __syncthreads();
        }
    }
    // ---- a += a1, a2 += a3, then a += a2 (sg_col_kernel's order): wavefront 2 folds wavefront 3's partials into its own, wavefront 0 folds
    // wavefront 1's and then wavefront 2's -- all through the LDS copies
    __syncthreads();
    Lse acc[17];
    auto fetch = [&](Lse (&d)[17], int rr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 pm = am4[rr][k][lane], ps = as4[rr][k][lane];
            d[4 * k] = { pm.x, ps.x }; d[4 * k + 1] = { pm.y, ps.y }; d[4 * k + 2] = { pm.z, ps.z }; d[4 * k + 3] = { pm.w, ps.w };
        }
        d[16] = { amd[rr][lane], asd[rr][lane] };
    };
    if (w8 == 2) {
        Lse o[17];
        fetch(acc, 2); fetch(o, 3);
#pragma unroll
        for (int k = 0; k < 17; ++k) lse_merge(acc[k], o[k].m, o[k].s);
#pragma unroll
        for (int k = 0; k < 17; ++k) { xm[0][k][lane] = acc[k].m; xs[0][k][lane] = acc[k].s; }
    } else if (w8 == 0) {
        Lse o[17];
        fetch(acc, 0); fetch(o, 1);
#pragma unroll
        for (int k = 0; k < 17; ++k) lse_merge(acc[k], o[k].m, o[k].s);
    }
    __syncthreads();
    if (w8 == 0) {
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
            hipLaunchKernelGGL(sg_sweep_kernel, dim3(16, B), dim3(512), 0, s, S, ldS, n0, n1, bin_score, v, u, part_m, part_s);
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
