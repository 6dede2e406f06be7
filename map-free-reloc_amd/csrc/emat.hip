// emat.hip -- essential-matrix path of the relative-pose solver on gfx950 (MI355X).
//
// Replaces EssentialMatrixSolver.estimate_pose (lib/models/matching/pose_solver.py:29-61) for a
// BATCH of image pairs:
//   emat_prep_kernel    K-normalise both views in f32 (:39-40), thr = PIX_THRESHOLD / mean f (:43, Q8)
//   emat_hyp_kernel     cv.findEssentialMat hypotheses: one LANE per 5-point minimal solve (Nister; <= 10 candidate E
//                       each), working set in LDS (emat_lds.h: 145 KB per 64-lane workgroup)       (:46-48)
//   emat_score_kernel   one WAVEFRONT per hypothesis: lanes stride over the LDS-staged correspondences, squared Sampson
//                       distance -> MAGSAC++ loss through the LDS-resident table (per-lane sums, wave butterfly per tile of
//                       1024 points) + tentative inliers r^2 < thr^2 by ballot + popcount; or (score COUNT) the count alone
//   emat_select_kernel  RANSAC replay (adaptive iteration cap; with MAGSAC: smallest loss wins and every new best model from
//                       iteration 100 on goes through sigma-consensus++ = IRLS with the MAGSAC++ weights on (R, unit t))
//                       -> best E -> cv.recoverPose (4 decompositions, cheirality vote; :56-60) -> final cheirality-filtered
//                       inlier mask (what self.mask holds, Q7).  Score COUNT: LM polish of (R, t) on the inliers instead.
// score MFR_EMAT_SCORE_MAGSAC is what the reference asks OpenCV for (method=cv.USAC_MAGSAC, pose_solver.py:46-48), restated
// from the MAGSAC++ paper + USAC's control flow; OpenCV's exact constants are not reproducible offline (parity unpinned vs
// OpenCV).  Bit-exact twin: oracle/mfr_oracle_emat.c.
// Compiled with -ffp-contract=off (FP contract in geom_dev.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "emat_dev.h"
#include "emat_lds.h"

using namespace mfr;

// hypotheses per workgroup of emat_score_kernel (4 wavefronts, 4 hypotheses each).  Round 5: 64 -> 16 -- at 1000 iterations x 16 pairs the grid
// was 256 workgroups = ONE wavefront per SIMD of latency-bound fp64 work; a hypothesis' result does not depend on the grouping
#define EM_HYP_PER_WG 16
#define EM_TILE 1024
#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

__global__ void __launch_bounds__(256) emat_prep_kernel(
    const float *__restrict__ pts0, const float *__restrict__ pts1, const int32_t *__restrict__ n_corr, int maxN,
    const void *__restrict__ K0, const void *__restrict__ K1, int k_dtype, double pix_thr,
    double *__restrict__ x0, double *__restrict__ x1, double *__restrict__ thr2)
{
    // pose_solver.py:39-43 in the dtype K arrives in: numpy float32 arithmetic for a float32 K, float64 (keypoints widened) for
    // the Map-free loader's float64 K; the threshold mean likewise
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    const size_t o = ((size_t)b * maxN + i) * 2;
    if (k_dtype == MFR_K_F32) {
        const float *k0 = (const float *)K0 + 9 * b, *k1 = (const float *)K1 + 9 * b;
        if (i == 0) {
            const float m = (((k0[0] + k1[4]) + k0[4]) + k1[0]) / 4.0f;       // np.mean([fx0, fy1, fy0, fx1]) in f32
            const double thr = pix_thr / (double)m;
            thr2[b] = thr * thr;
        }
        if (i >= n) return;
        x0[o] = (double)((pts0[o] - k0[2]) / k0[0]); x0[o + 1] = (double)((pts0[o + 1] - k0[5]) / k0[4]);
        x1[o] = (double)((pts1[o] - k1[2]) / k1[0]); x1[o + 1] = (double)((pts1[o + 1] - k1[5]) / k1[4]);
    } else {
        const double *k0 = (const double *)K0 + 9 * b, *k1 = (const double *)K1 + 9 * b;
        if (i == 0) {
            const double m = (((k0[0] + k1[4]) + k0[4]) + k1[0]) / 4.0;
            const double thr = pix_thr / m;
            thr2[b] = thr * thr;
        }
        if (i >= n) return;
        x0[o] = ((double)pts0[o] - k0[2]) / k0[0]; x0[o + 1] = ((double)pts0[o + 1] - k0[5]) / k0[4];
        x1[o] = ((double)pts1[o] - k1[2]) / k1[0]; x1[o + 1] = ((double)pts1[o + 1] - k1[5]) / k1[4];
    }
}

// grid (ceil(iters/64), B), one lane per hypothesis
__global__ void __launch_bounds__(64) emat_hyp_kernel(
    const double *__restrict__ x0, const double *__restrict__ x1, const int32_t *__restrict__ n_corr, int maxN,
    int max_iters, uint64_t seed, const int64_t *__restrict__ pair_ids, double *__restrict__ Es /*[B,iters,10,9]*/,
    int32_t *__restrict__ nsol /*[B,iters]*/)
{
    // the solver's working set (Gauss-Jordan tableaux, monomial tables, derivative stack) in LDS, lane-interleaved: emat_lds.h
    __shared__ double fp_lds[FP_LDS_DOUBLES * 64];
    __shared__ int fp_colp[9 * 64];
    const int b = blockIdx.y, it = blockIdx.x * 64 + threadIdx.x;
    if (it >= max_iters) return;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    int ns = 0;
    double *o = Es + ((size_t)b * max_iters + it) * 90;
    if (n >= 5 && (n > 5 || it == 0)) {
        int s[5];
        if (n == 5) {
#pragma unroll
            for (int k = 0; k < 5; ++k) s[k] = k;
        }
        else sample_distinct<5>(seed, (uint64_t)pair_ids[b], (uint32_t)it, n, s);
        double a[10], c[10];
        const double *p0 = x0 + (size_t)b * maxN * 2, *p1 = x1 + (size_t)b * maxN * 2;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            a[2 * k] = p0[2 * s[k]]; a[2 * k + 1] = p0[2 * s[k] + 1];
            c[2 * k] = p1[2 * s[k]]; c[2 * k + 1] = p1[2 * s[k] + 1];
        }
        ns = fivept_lds(a, c, o, fp_lds + threadIdx.x, fp_colp + threadIdx.x, true);      // -1: front end done, 86 doubles in the slot
    }
    nsol[(size_t)b * max_iters + it] = ns;
}

// root stage: 8 lanes per hypothesis, 32 hypotheses per workgroup (emat_lds.h fp_roots_group); turns the 86-double front-end
// record of a hypothesis into its <= 10 essential matrices in place
__global__ void __launch_bounds__(256) emat_roots_kernel(double *__restrict__ Es, int32_t *__restrict__ nsol, int total)
{
    __shared__ FprShared sh[FPR_HYP_PER_WG];
    const int g = threadIdx.x / FPR_GROUP, sub = threadIdx.x % FPR_GROUP;
    const int h = blockIdx.x * FPR_HYP_PER_WG + g;
    const bool active = h < total && nsol[h < total ? h : 0] == -1;
    fp_roots_group(Es + (size_t)(h < total ? h : 0) * 90, nsol + (h < total ? h : 0), &sh[g], sub, active);
}

#define MAGSAC_LO_START 100      // USAC's max_iters_before_LO
#define MAGSAC_LO_ITERS 20       // re-weighting rounds per local optimisation
#define MAGSAC_MAX_M 2048        // table intervals the LDS images are sized for

// normalised MAGSAC++ loss of a squared residual below the cut: linear interpolation in the table (entry stride `st` doubles)
static __device__ __forceinline__ double magsac_interp(const double *lut, int st, int M, double scale, double r2)
{
    const double u = r2 * scale;
    int j = (int)u;
    if (j > M - 1) j = M - 1;
    const double f = u - (double)j;
    const double a = lut[st * j], b = lut[st * (j + 1)];
    return a + f * (b - a);
}

// grid (ceil(iters / EM_HYP_PER_WG), B), 4 wavefronts; wavefront w scores hypotheses w, w+4, ... of the block's EM_HYP_PER_WG
template <bool MAGSAC>
__global__ void __launch_bounds__(256) emat_score_kernel(
    const double *__restrict__ x0, const double *__restrict__ x1, const int32_t *__restrict__ n_corr, int maxN,
    int max_iters, const double *__restrict__ thr2p, const double *__restrict__ Es, const int32_t *__restrict__ nsol,
    const double *__restrict__ lut, int lut_m, double ratio2,
    int32_t *__restrict__ counts /*[B,iters]*/, int32_t *__restrict__ bestm /*[B,iters]*/, double *__restrict__ losses /*[B,iters]*/)
{
    __shared__ double ta[EM_TILE], tb[EM_TILE], tc[EM_TILE], td[EM_TILE];
    __shared__ int cnt[EM_HYP_PER_WG][10];
    __shared__ double lsum[MAGSAC ? EM_HYP_PER_WG : 1][10];
    __shared__ double lutl[MAGSAC ? MAGSAC_MAX_M + 1 : 1];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    const double thr2 = thr2p[b];
    const double cut = ratio2 * thr2, scale = (double)lut_m / cut;
    const int it0 = blockIdx.x * EM_HYP_PER_WG;
    for (int i = tid; i < EM_HYP_PER_WG * 10; i += 256) {
        cnt[i / 10][i % 10] = 0;
        if (MAGSAC) lsum[i / 10][i % 10] = 0.0;
    }
    if (MAGSAC)
        for (int i = tid; i <= lut_m; i += 256) lutl[i] = lut[2 * i];
    const double *p0 = x0 + (size_t)b * maxN * 2, *p1 = x1 + (size_t)b * maxN * 2;
    for (int base = 0; base < n; base += EM_TILE) {
        const int tn = min(EM_TILE, n - base);
        __syncthreads();
        for (int i = tid; i < tn; i += 256) {
            ta[i] = p0[2 * (size_t)(base + i)]; tb[i] = p0[2 * (size_t)(base + i) + 1];
            tc[i] = p1[2 * (size_t)(base + i)]; td[i] = p1[2 * (size_t)(base + i) + 1];
        }
        __syncthreads();
        for (int h = wid; h < EM_HYP_PER_WG; h += 4) {
            const int it = it0 + h;
            if (it >= max_iters) break;
            const int ns = nsol[(size_t)b * max_iters + it];
            const double *Eh = Es + ((size_t)b * max_iters + it) * 90;
            for (int m = 0; m < ns; ++m) {
                double E[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) E[k] = Eh[9 * m + k];
                int c = 0;
                double acc = 0.0;
                for (int i0 = 0; i0 < tn; i0 += 64) {
                    const int i = i0 + lane;
                    bool in = false;
                    if (i < tn) {
                        const double r2 = sampson2(E, ta[i], tb[i], tc[i], td[i]);
                        if (MAGSAC) {
                            in = r2 < thr2;
                            if (r2 < cut) acc = acc + magsac_interp(lutl, 1, lut_m, scale, r2);
                        } else in = r2 <= thr2;
                    }
                    c += __popcll(__ballot(in));
                }
                if (MAGSAC) acc = wave_sum(acc);
                if (lane == 0) {
                    cnt[h][m] += c;
                    if (MAGSAC) lsum[h][m] = lsum[h][m] + acc;        // tiles added in sequence (the oracle's order)
                }
            }
        }
    }
    __syncthreads();
    if (tid < EM_HYP_PER_WG) {
        const int it = it0 + tid;
        if (it < max_iters) {
            const int ns = nsol[(size_t)b * max_iters + it];
            int best = 0, bm = -1;
            if (MAGSAC) {
                double bl = 0.0;
                for (int m = 0; m < ns; ++m) if (lsum[tid][m] < bl) { bl = lsum[tid][m]; best = cnt[tid][m]; bm = m; }   // first min
                losses[(size_t)b * max_iters + it] = bl;
            } else {
                for (int m = 0; m < ns; ++m) if (cnt[tid][m] > best) { best = cnt[tid][m]; bm = m; }   // first max
            }
            counts[(size_t)b * max_iters + it] = best;
            bestm[(size_t)b * max_iters + it] = bm;
        }
    }
}

// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ void quat_right_update(const double *R, const double *dw, double *Rn)
{
    const double hx = 0.5 * dw[0], hy = 0.5 * dw[1], hz = 0.5 * dw[2];
    const double nn = sqrt(((hx * hx + hy * hy) + hz * hz) + 1.0);
    const double w = 1.0 / nn, x = hx / nn, y = hy / nn, z = hz / nn;
    double Q[9];
    Q[0] = 1.0 - 2.0 * (y * y + z * z); Q[1] = 2.0 * (x * y - w * z);       Q[2] = 2.0 * (x * z + w * y);
    Q[3] = 2.0 * (x * y + w * z);       Q[4] = 1.0 - 2.0 * (x * x + z * z); Q[5] = 2.0 * (y * z - w * x);
    Q[6] = 2.0 * (x * z - w * y);       Q[7] = 2.0 * (y * z + w * x);       Q[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = (R[3 * i] * Q[j] + R[3 * i + 1] * Q[3 + j]) + R[3 * i + 2] * Q[6 + j];
}

static __device__ __forceinline__ int chol_solve6(const double *A, const double *bvec, double *x)
{
    double L[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = s - L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0.0)) return -1;
                L[6 * i + i] = sqrt(s);
            } else L[6 * i + j] = s / L[6 * j + j];
        }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = bvec[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = s - L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s = s - L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return 0;
}

// emat_select_kernel runs one workgroup of EM_SEL_WAVES wavefronts per pair (round 5; rounds 1-4: ONE wavefront per pair -- 16 wavefronts on the
// whole GPU for LoFTR's 16 pairs, 1.85 ms of pure latency).  Sums over points: point i belongs to thread i mod (64 EM_SEL_WAVES), a wavefront's
// 64 partials merge in the xor butterfly (wave_sum), the wavefront totals are added in sequence -- the order oracle/mfr_oracle_emat.c's
// wacc_finish restates.  Every thread ends up with the same value, so all control flow stays uniform over the workgroup.
// 4 wavefronts: the kernel needs ~340 registers per lane (the 6 x 6 normal equations and their Cholesky factor are register-resident);
// 8 wavefronts would cap it at 256 and spill 1.5 KB per lane
#define EM_SEL_WAVES 4
#define EM_SEL_THREADS (64 * EM_SEL_WAVES)
struct SelRed { double d[EM_SEL_WAVES][27]; int i[EM_SEL_WAVES]; };
template <int N>
static __device__ __forceinline__ void block_sum(SelRed &rd, double (&acc)[N])
{
    const int wv = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = wave_sum(acc[q]);
    if (lane_id() == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) rd.d[wv][q] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double s = rd.d[0][q];
#pragma unroll
        for (int w = 1; w < EM_SEL_WAVES; ++w) s = s + rd.d[w][q];
        acc[q] = s;
    }
    __syncthreads();
}
static __device__ __forceinline__ int block_isum(SelRed &rd, int c)      // c: wave-uniform partial count
{
    if (lane_id() == 0) rd.i[threadIdx.x >> 6] = c;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < EM_SEL_WAVES; ++w) s += rd.i[w];
    __syncthreads();
    return s;
}

static __device__ __forceinline__ double emat_cost(SelRed &rd, const double *p0, const double *p1, const int32_t *idx, int n,
                                                   const double *R, const double *t)
{
    double E[9];
    skew_mul(t, R, E);
    double acc[1] = { 0.0 };
    for (int i = (int)threadIdx.x; i < n; i += EM_SEL_THREADS) {
        const int j = idx[i];
        acc[0] = acc[0] + sampson2(E, p0[2 * (size_t)j], p0[2 * (size_t)j + 1], p1[2 * (size_t)j], p1[2 * (size_t)j + 1]);
    }
    block_sum(rd, acc);
    return acc[0];
}

// LM polish of (R, unit t) on the Sampson cost (workgroup-parallel, block_sum-ordered reductions)
static __device__ __forceinline__ int emat_refine(SelRed &rd, const double *p0, const double *p1, const int32_t *idx, int n,
                                               int max_iter, double *R, double *t)
{
    double lambda = 1e-3;
    double cost = emat_cost(rd, p0, p1, idx, n, R, t);
    if (!(cost == cost)) return -1;
    for (int it = 0; it < max_iter; ++it) {
        double E[9];
        skew_mul(t, R, E);
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        for (int i = (int)threadIdx.x; i < n; i += EM_SEL_THREADS) {
            const int j = idx[i];
            const double a = p0[2 * (size_t)j], b = p0[2 * (size_t)j + 1], c = p1[2 * (size_t)j], d = p1[2 * (size_t)j + 1];
            const double Ex0 = (E[0] * a + E[1] * b) + E[2], Ex1 = (E[3] * a + E[4] * b) + E[5], Ex2 = (E[6] * a + E[7] * b) + E[8];
            const double Et0 = (E[0] * c + E[3] * d) + E[6], Et1 = (E[1] * c + E[4] * d) + E[7];
            const double num = (c * Ex0 + d * Ex1) + Ex2;
            const double den = ((Ex0 * Ex0 + Ex1 * Ex1) + Et0 * Et0) + Et1 * Et1;
            const double wgt = 1.0 / sqrt(den);
            const double q[3] = { c, d, 1.0 }, p[3] = { a, b, 1.0 };
            const double txq[3] = { t[1] * q[2] - t[2] * q[1], t[2] * q[0] - t[0] * q[2], t[0] * q[1] - t[1] * q[0] };
            const double u[3] = { -((R[0] * txq[0] + R[3] * txq[1]) + R[6] * txq[2]),
                                  -((R[1] * txq[0] + R[4] * txq[1]) + R[7] * txq[2]),
                                  -((R[2] * txq[0] + R[5] * txq[1]) + R[8] * txq[2]) };
            const double Rp[3] = { (R[0] * p[0] + R[1] * p[1]) + R[2] * p[2], (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2],
                                   (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2] };
            double J[6];
            J[0] = (p[1] * u[2] - p[2] * u[1]) * wgt; J[1] = (p[2] * u[0] - p[0] * u[2]) * wgt; J[2] = (p[0] * u[1] - p[1] * u[0]) * wgt;
            J[3] = (Rp[1] * q[2] - Rp[2] * q[1]) * wgt; J[4] = (Rp[2] * q[0] - Rp[0] * q[2]) * wgt; J[5] = (Rp[0] * q[1] - Rp[1] * q[0]) * wgt;
            const double r = num * wgt;
            int qq = 0;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr)
#pragma unroll
                for (int cc = rr; cc < 6; ++cc, ++qq) acc[qq] = acc[qq] + J[rr] * J[cc];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr, ++qq) acc[qq] = acc[qq] + J[rr] * r;
        }
        block_sum(rd, acc);
        double H[36], g[6];
        {
            int qq = 0;
            for (int rr = 0; rr < 6; ++rr)
                for (int cc = rr; cc < 6; ++cc, ++qq) { H[6 * rr + cc] = acc[qq]; H[6 * cc + rr] = acc[qq]; }
            for (int rr = 0; rr < 6; ++rr, ++qq) g[rr] = -acc[qq];
        }
        for (int rr = 0; rr < 6; ++rr) H[6 * rr + rr] = H[6 * rr + rr] + lambda * H[6 * rr + rr];
        for (int rr = 0; rr < 3; ++rr)
            for (int cc = 0; cc < 3; ++cc) H[6 * (3 + rr) + 3 + cc] = H[6 * (3 + rr) + 3 + cc] + t[rr] * t[cc];
        double dl[6];
        if (chol_solve6(H, g, dl)) {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
            continue;
        }
        double Rn[9], tn[3];
        quat_right_update(R, dl, Rn);
        tn[0] = t[0] + dl[3]; tn[1] = t[1] + dl[4]; tn[2] = t[2] + dl[5];
        const double nt = sqrt((tn[0] * tn[0] + tn[1] * tn[1]) + tn[2] * tn[2]);
        if (!(nt > 0.0)) {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
            continue;
        }
        tn[0] = tn[0] / nt; tn[1] = tn[1] / nt; tn[2] = tn[2] / nt;
        const double cn = emat_cost(rd, p0, p1, idx, n, Rn, tn);
        double mx = 0.0;
        for (int k = 0; k < 6; ++k) { const double v = dl[k] < 0.0 ? -dl[k] : dl[k]; if (v > mx) mx = v; }
        if (cn < cost) {
            const double dec = cost - cn;
            for (int k = 0; k < 9; ++k) R[k] = Rn[k];
            for (int k = 0; k < 3; ++k) t[k] = tn[k];
            const bool done = (dec <= 1e-14 * cost);
            cost = cn;
            lambda = lambda * 0.1;
            if (lambda < 1e-12) lambda = 1e-12;
            if (done) break;
        } else {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
        }
        if (mx < 1e-13) break;
    }
    return 0;
}

// workgroup-wide count of a per-point predicate over all n points
template <typename F>
static __device__ __forceinline__ int block_count(SelRed &rd, int n, F pred)
{
    int c = 0;
    for (int i0 = 0; i0 < n; i0 += EM_SEL_THREADS) {
        const int i = i0 + (int)threadIdx.x;
        c += __popcll(__ballot(i < n && pred(i)));
    }
    return block_isum(rd, c);
}

// two Newton-Schulz steps towards the orthogonal polar factor, R <- R (3 I - R^T R) / 2: Horn's closed-form R inherits E's distance
// from the essential manifold (~1e-9 for a five-point solution); after this it is a rotation up to rounding
static __device__ __forceinline__ void orthonormalize(double *R)
{
    for (int it = 0; it < 2; ++it) {
        double S[9], Rn[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double g = (R[i] * R[j] + R[3 + i] * R[3 + j]) + R[6 + i] * R[6 + j];
                S[3 * i + j] = ((i == j) ? 3.0 : 0.0) - g;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Rn[3 * i + j] = 0.5 * ((R[3 * i] * S[j] + R[3 * i + 1] * S[3 + j]) + R[3 * i + 2] * S[6 + j]);
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
    }
}

struct Magsac { const double *lut; int M; double cut, scale, thr2; };     // lut: LDS image, entry j = (loss, weight)

// total MAGSAC++ loss + tentative inlier count of one model, one workgroup: a wavefront takes a tile of 1024 points (per-lane sums, butterfly),
// EM_SEL_WAVES tiles at a time, and the tile totals are added in sequence (the order emat_score_kernel and the oracle use)
static __device__ __forceinline__ double magsac_score_block(SelRed &rd, const Magsac &ms, const double *E, const double *p0, const double *p1, int n,
                                                            int *cnt_out)
{
    double L = 0.0;
    int c = 0;
    const int wv = (int)(threadIdx.x >> 6);
    const int ntiles = (n + EM_TILE - 1) / EM_TILE;
    for (int tb = 0; tb < ntiles; tb += EM_SEL_WAVES) {
        const int base = (tb + wv) * EM_TILE;
        const int tn = min(EM_TILE, n - base);               // <= 0: no tile for this wavefront in this round
        double acc = 0.0;
        int cw = 0;
        for (int i0 = 0; i0 < tn; i0 += 64) {
            const int i = i0 + lane_id();
            bool in = false;
            if (i < tn) {
                const size_t j = (size_t)(base + i);
                const double r2 = sampson2(E, p0[2 * j], p0[2 * j + 1], p1[2 * j], p1[2 * j + 1]);
                in = r2 < ms.thr2;
                if (r2 < ms.cut) acc = acc + magsac_interp(ms.lut, 2, ms.M, ms.scale, r2);
            }
            cw += __popcll(__ballot(in));
        }
        acc = wave_sum(acc);
        if (lane_id() == 0) { rd.d[wv][0] = acc; rd.i[wv] = cw; }
        __syncthreads();
        for (int w = 0; w < EM_SEL_WAVES && tb + w < ntiles; ++w) { L = L + rd.d[w][0]; c += rd.i[w]; }
        __syncthreads();
    }
    *cnt_out = c;
    return L;
}

// sigma-consensus++: iteratively re-weighted least squares with the MAGSAC++ weights (= d loss / d r^2) on (R, unit t); one damped
// Gauss-Newton step per re-weighting round, accepted when the MAGSAC++ loss decreases.  0 + optimised E / loss / count, or -1.
static __device__ __forceinline__ int magsac_lo_block(SelRed &rd, const Magsac &ms, const double *p0, const double *p1, int n, const double *Ein,
                                                  double *Eout, double *loss_out, int *cnt_out)
{
    double R[9], Rb[9], t[3];
    if (emat_decompose(Ein, R, Rb, t)) return -1;
    orthonormalize(R);
    double E[9];
    skew_mul(t, R, E);
    int cnt;
    double loss = magsac_score_block(rd, ms, E, p0, p1, n, &cnt);
    if (!(loss == loss)) return -1;
    double lambda = 1e-3;
    for (int it = 0; it < MAGSAC_LO_ITERS; ++it) {
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        for (int i = (int)threadIdx.x; i < n; i += EM_SEL_THREADS) {
            const double a = p0[2 * (size_t)i], b = p0[2 * (size_t)i + 1], c = p1[2 * (size_t)i], d = p1[2 * (size_t)i + 1];
            const double Ex0 = (E[0] * a + E[1] * b) + E[2], Ex1 = (E[3] * a + E[4] * b) + E[5], Ex2 = (E[6] * a + E[7] * b) + E[8];
            const double Et0 = (E[0] * c + E[3] * d) + E[6], Et1 = (E[1] * c + E[4] * d) + E[7];
            const double num = (c * Ex0 + d * Ex1) + Ex2;
            const double den = ((Ex0 * Ex0 + Ex1 * Ex1) + Et0 * Et0) + Et1 * Et1;
            const double r2 = (num * num) / den;
            if (!(r2 < ms.cut)) continue;
            const double pw = magsac_interp(ms.lut + 1, 2, ms.M, ms.scale, r2);
            const double wgt = 1.0 / sqrt(den);
            const double q[3] = { c, d, 1.0 }, p[3] = { a, b, 1.0 };
            const double txq[3] = { t[1] * q[2] - t[2] * q[1], t[2] * q[0] - t[0] * q[2], t[0] * q[1] - t[1] * q[0] };
            const double u[3] = { -((R[0] * txq[0] + R[3] * txq[1]) + R[6] * txq[2]),
                                  -((R[1] * txq[0] + R[4] * txq[1]) + R[7] * txq[2]),
                                  -((R[2] * txq[0] + R[5] * txq[1]) + R[8] * txq[2]) };
            const double Rp[3] = { (R[0] * p[0] + R[1] * p[1]) + R[2] * p[2], (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2],
                                   (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2] };
            double J[6];
            J[0] = (p[1] * u[2] - p[2] * u[1]) * wgt; J[1] = (p[2] * u[0] - p[0] * u[2]) * wgt; J[2] = (p[0] * u[1] - p[1] * u[0]) * wgt;
            J[3] = (Rp[1] * q[2] - Rp[2] * q[1]) * wgt; J[4] = (Rp[2] * q[0] - Rp[0] * q[2]) * wgt; J[5] = (Rp[0] * q[1] - Rp[1] * q[0]) * wgt;
            const double r = num * wgt;
            int qq = 0;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) {
                const double wj = pw * J[rr];
#pragma unroll
                for (int cc = rr; cc < 6; ++cc, ++qq) acc[qq] = acc[qq] + wj * J[cc];
            }
#pragma unroll
            for (int rr = 0; rr < 6; ++rr, ++qq) acc[qq] = acc[qq] + (pw * J[rr]) * r;
        }
        block_sum(rd, acc);
        double H[36], g[6];
        {
            int qq = 0;
            for (int rr = 0; rr < 6; ++rr)
                for (int cc = rr; cc < 6; ++cc, ++qq) { H[6 * rr + cc] = acc[qq]; H[6 * cc + rr] = acc[qq]; }
            for (int rr = 0; rr < 6; ++rr, ++qq) g[rr] = -acc[qq];
        }
        for (int rr = 0; rr < 6; ++rr) H[6 * rr + rr] = H[6 * rr + rr] + lambda * H[6 * rr + rr];
        for (int rr = 0; rr < 3; ++rr)
            for (int cc = 0; cc < 3; ++cc) H[6 * (3 + rr) + 3 + cc] = H[6 * (3 + rr) + 3 + cc] + t[rr] * t[cc];
        double dl[6];
        if (chol_solve6(H, g, dl)) {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
            continue;
        }
        double Rn[9], tn[3], En[9];
        quat_right_update(R, dl, Rn);
        tn[0] = t[0] + dl[3]; tn[1] = t[1] + dl[4]; tn[2] = t[2] + dl[5];
        const double nt = sqrt((tn[0] * tn[0] + tn[1] * tn[1]) + tn[2] * tn[2]);
        if (!(nt > 0.0)) {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
            continue;
        }
        tn[0] = tn[0] / nt; tn[1] = tn[1] / nt; tn[2] = tn[2] / nt;
        skew_mul(tn, Rn, En);
        int cn;
        const double ln = magsac_score_block(rd, ms, En, p0, p1, n, &cn);
        double mx = 0.0;
        for (int k = 0; k < 6; ++k) { const double v = dl[k] < 0.0 ? -dl[k] : dl[k]; if (v > mx) mx = v; }
        if (ln < loss) {
            const double dec = loss - ln, mag = loss < 0.0 ? -loss : loss;
            for (int k = 0; k < 9; ++k) { R[k] = Rn[k]; E[k] = En[k]; }
            for (int k = 0; k < 3; ++k) t[k] = tn[k];
            const bool done = (dec <= 1e-12 * mag);
            loss = ln; cnt = cn;
            lambda = lambda * 0.1;
            if (lambda < 1e-12) lambda = 1e-12;
            if (done) break;
        } else {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
        }
        if (mx < 1e-13) break;
    }
    for (int k = 0; k < 9; ++k) Eout[k] = E[k];
    *loss_out = loss; *cnt_out = cnt;
    return 0;
}

template <bool MAGSAC>
__global__ void __launch_bounds__(EM_SEL_THREADS) emat_select_kernel(
    const double *__restrict__ x0, const double *__restrict__ x1, const int32_t *__restrict__ n_corr, int maxN,
    int max_iters, const double *__restrict__ thr2p, double conf, const double *__restrict__ Es,
    const int32_t *__restrict__ counts, const int32_t *__restrict__ bestm, const double *__restrict__ losses,
    const double *__restrict__ lut, int lut_m, double ratio2, int32_t *__restrict__ idx_ws /*[B,maxN]*/,
    uint8_t *__restrict__ rmask_ws /*[B,maxN]*/, double *__restrict__ Rout, double *__restrict__ tout,
    int32_t *__restrict__ n_inliers, int32_t *__restrict__ status, uint8_t *__restrict__ mask_out,
    int32_t *__restrict__ best_iter, int32_t *__restrict__ iters_run, int32_t *__restrict__ lo_runs)
{
    __shared__ double lutl[MAGSAC ? 2 * (MAGSAC_MAX_M + 1) : 1];
    __shared__ SelRed rd;
    const int b = blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    const double *p0 = x0 + (size_t)b * maxN * 2, *p1 = x1 + (size_t)b * maxN * 2;
    const double thr2 = thr2p[b];
    int32_t *idx = idx_ws + (size_t)b * maxN;
    uint8_t *rm = rmask_ws + (size_t)b * maxN;
    uint8_t *mo = mask_out ? mask_out + (size_t)b * maxN : nullptr;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    if (mo) for (int i = tid; i < maxN; i += EM_SEL_THREADS) mo[i] = 0;
    Magsac ms;
    if (MAGSAC) {
        for (int i = tid; i < 2 * (lut_m + 1); i += EM_SEL_THREADS) lutl[i] = lut[i];
        __syncthreads();
        ms.lut = lutl; ms.M = lut_m; ms.thr2 = thr2; ms.cut = ratio2 * thr2; ms.scale = (double)lut_m / ms.cut;
    }
    int st = (n < 5) ? MFR_ST_TOO_FEW : MFR_ST_OK;
    int bit = -1, best = MAGSAC ? 0 : 4, run = 0, nlo = 0;
    bool best_is_lo = false;
    double best_loss = 0.0;
    double Eb[9];
    const int32_t *cnt = counts + (size_t)b * max_iters;
    if (st == MFR_ST_OK && MAGSAC) {
        const double *ls = losses + (size_t)b * max_iters;
        if (n == 5) {
            run = 1;
            if (ls[0] < 0.0) { best = cnt[0]; bit = 0; best_loss = ls[0]; }
        } else {
            // (every wavefront replays the loop on the same values: uniform control flow over the workgroup, the local optimisation inside is
            // workgroup-parallel)
            // replay of the sequential loop: the next hypothesis that beats the best loss so far becomes the best model, goes through
            // the local optimisation (from iteration MAGSAC_LO_START on) and lowers the iteration cap
            int niters = max_iters;
            for (int c0 = 0; c0 < max_iters && c0 < niters; c0 += 64) {
                const int it = c0 + lane;
                const double v = (it < max_iters) ? ls[it] : 0.0;
                int from = 0;
                while (true) {
                    const unsigned long long rec = __ballot(v < best_loss && lane >= from && it < niters);
                    if (!rec) break;
                    const int l = __ffsll((long long)rec) - 1;
                    const int itr = c0 + l;
                    best_loss = __shfl(v, l, 64);
                    best = cnt[itr];
                    bit = itr;
                    best_is_lo = false;
                    if (itr >= MAGSAC_LO_START) {
                        const double *src = Es + ((size_t)b * max_iters + itr) * 90 + 9 * bestm[(size_t)b * max_iters + itr];
                        double Ein[9], El[9], ll;
                        int cl;
                        for (int k = 0; k < 9; ++k) Ein[k] = src[k];
                        ++nlo;
                        if (magsac_lo_block(rd, ms, p0, p1, n, Ein, El, &ll, &cl) == 0 && ll < best_loss) {
                            for (int k = 0; k < 9; ++k) Eb[k] = El[k];
                            best_loss = ll; best = cl; best_is_lo = true;
                        }
                    }
                    niters = update_num_iters(conf, (double)(n - best) / (double)n, 5, niters);
                    from = l + 1;
                }
            }
            run = (bit + 1 > niters) ? bit + 1 : niters;
        }
        if (bit < 0) st = MFR_ST_NO_MODEL;
    } else if (st == MFR_ST_OK) {
        if (n == 5) {
            run = 1;
            if (cnt[0] > 0) { best = cnt[0]; bit = 0; }
        } else {
            int niters = max_iters, carry = 4;
            bool stop = false;
            for (int c0 = 0; c0 < max_iters && !stop && c0 < niters; c0 += 64) {
                const int it = c0 + lane;
                const int v = (it < max_iters) ? cnt[it] : -1;
                int incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off, 64);
                    if (lane >= off && o > incl) incl = o;
                }
                int excl = __shfl_up(incl, 1, 64);
                if (lane == 0 || excl < carry) excl = carry;
                unsigned long long rec = __ballot(v > excl);
                while (rec) {
                    const int l = __ffsll((long long)rec) - 1;
                    rec &= rec - 1;
                    const int itr = c0 + l;
                    if (itr >= niters) { stop = true; break; }
                    best = __shfl(v, l, 64);
                    bit = itr;
                    niters = update_num_iters(conf, (double)(n - best) / (double)n, 5, niters);
                }
                const int last = __shfl(incl, 63, 64);
                if (last > carry) carry = last;
            }
            run = (bit + 1 > niters) ? bit + 1 : niters;
        }
        if (bit < 0) st = MFR_ST_NO_MODEL;
    }
    double R[9], t[3];
    int m = 0;
    if (st == MFR_ST_OK) {
        if (!(MAGSAC && best_is_lo)) {
            const double *src = Es + ((size_t)b * max_iters + bit) * 90 + 9 * bestm[(size_t)b * max_iters + bit];
            for (int k = 0; k < 9; ++k) Eb[k] = src[k];
        }
        if (MAGSAC && !best_is_lo && n > 5) {        // the winner never went through the local optimisation
            double El[9], ll;
            int cl;
            ++nlo;
            if (magsac_lo_block(rd, ms, p0, p1, n, Eb, El, &ll, &cl) == 0 && ll < best_loss) {
                for (int k = 0; k < 9; ++k) Eb[k] = El[k];
                best_loss = ll; best = cl;
            }
        }
        // RANSAC inlier set of the best model (ascending index order); USAC compares strictly.  Ordered compaction: wavefront 0 alone
        if (wv == 0) {
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                bool in = false;
                if (i < n) {
                    const double r2 = sampson2(Eb, p0[2 * (size_t)i], p0[2 * (size_t)i + 1], p1[2 * (size_t)i], p1[2 * (size_t)i + 1]);
                    in = MAGSAC ? (r2 < thr2) : (r2 <= thr2);
                }
                const unsigned long long bal = __ballot(in);
                if (in) idx[m + __popcll(bal & ((1ull << lane) - 1ull))] = i;
                if (i < n) rm[i] = in ? 1 : 0;
                m += __popcll(bal);
            }
            if (lane == 0) rd.i[0] = m;
        }
        __threadfence();
        __syncthreads();
        m = rd.i[0];
        __syncthreads();
        double Ra[9], Rc[9], tu[3];
        if (emat_decompose(Eb, Ra, Rc, tu)) st = MFR_ST_NO_MODEL;
        else {
            int bestc = -1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double *Rk = (c < 2) ? Ra : Rc;
                const double tk[3] = { (c & 1) ? -tu[0] : tu[0], (c & 1) ? -tu[1] : tu[1], (c & 1) ? -tu[2] : tu[2] };
                const int cc = block_count(rd, m, [&](int q) {
                    const int i = idx[q];
                    return cheirality(Rk, tk, p0[2 * (size_t)i], p0[2 * (size_t)i + 1], p1[2 * (size_t)i], p1[2 * (size_t)i + 1]);
                });
                if (cc > bestc) {
                    bestc = cc;
                    for (int k = 0; k < 9; ++k) R[k] = Rk[k];
                    for (int k = 0; k < 3; ++k) t[k] = tk[k];
                }
            }
            if (bestc <= 0) st = MFR_ST_NO_MODEL;
        }
        if (!MAGSAC && st == MFR_ST_OK && n > 5) {
            double Rr[9], tr[3];
            for (int k = 0; k < 9; ++k) Rr[k] = R[k];
            for (int k = 0; k < 3; ++k) tr[k] = t[k];
            if (emat_refine(rd, p0, p1, idx, m, 20, Rr, tr) == 0) {
                double Er[9];
                skew_mul(tr, Rr, Er);
                const int m2 = block_count(rd, n, [&](int i) {
                    return sampson2(Er, p0[2 * (size_t)i], p0[2 * (size_t)i + 1], p1[2 * (size_t)i], p1[2 * (size_t)i + 1]) <= thr2;
                });
                if (m2 >= m) {
                    for (int k = 0; k < 9; ++k) R[k] = Rr[k];
                    for (int k = 0; k < 3; ++k) t[k] = tr[k];
                    for (int i = tid; i < n; i += EM_SEL_THREADS)
                        rm[i] = (sampson2(Er, p0[2 * (size_t)i], p0[2 * (size_t)i + 1], p1[2 * (size_t)i], p1[2 * (size_t)i + 1]) <= thr2) ? 1 : 0;
                }
            }
        }
    }
    int cntf = 0;
    if (st == MFR_ST_OK) {
        orthonormalize(R);
        __threadfence();
        __syncthreads();                                    // rm: written by other wavefronts
        int cw = 0;
        for (int i0 = 0; i0 < n; i0 += EM_SEL_THREADS) {
            const int i = i0 + tid;
            bool in = false;
            if (i < n) in = rm[i] && cheirality(R, t, p0[2 * (size_t)i], p0[2 * (size_t)i + 1], p1[2 * (size_t)i], p1[2 * (size_t)i + 1]);
            if (mo && i < n) mo[i] = in ? 1 : 0;
            cw += __popcll(__ballot(in));
        }
        cntf = block_isum(rd, cw);
        if (cntf <= 0) st = MFR_ST_NO_MODEL;
    }
    if (st != MFR_ST_OK && mo) {
        __threadfence();
        __syncthreads();
        for (int i = tid; i < maxN; i += EM_SEL_THREADS) mo[i] = 0;
    }
    if (tid == 0) {
        for (int k = 0; k < 9; ++k) Rout[9 * b + k] = (st == MFR_ST_OK) ? R[k] : qnan;
        for (int k = 0; k < 3; ++k) tout[3 * b + k] = (st == MFR_ST_OK) ? t[k] : qnan;
        n_inliers[b] = (st == MFR_ST_OK) ? cntf : 0;
        status[b] = st;
        if (best_iter) best_iter[b] = bit;
        if (iters_run) iters_run[b] = run;
        if (lo_runs) lo_runs[b] = nlo;
    }
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
struct EmWs { size_t x0, x1, thr2, Es, nsol, counts, bestm, losses, idx, rm, total; };
static EmWs em_ws_layout(int B, int maxN, int iters)
{
    EmWs w; size_t o = 0;
    w.x0 = o;     o = align_up(o + sizeof(double) * 2 * (size_t)B * maxN, 256);
    w.x1 = o;     o = align_up(o + sizeof(double) * 2 * (size_t)B * maxN, 256);
    w.thr2 = o;   o = align_up(o + sizeof(double) * (size_t)B, 256);
    w.Es = o;     o = align_up(o + sizeof(double) * 90 * (size_t)B * iters, 256);
    w.nsol = o;   o = align_up(o + sizeof(int32_t) * (size_t)B * iters, 256);
    w.counts = o; o = align_up(o + sizeof(int32_t) * (size_t)B * iters, 256);
    w.bestm = o;  o = align_up(o + sizeof(int32_t) * (size_t)B * iters, 256);
    w.losses = o; o = align_up(o + sizeof(double) * (size_t)B * iters, 256);
    w.idx = o;    o = align_up(o + sizeof(int32_t) * (size_t)B * maxN, 256);
    w.rm = o;     o = align_up(o + (size_t)B * maxN, 256);
    w.total = o;
    return w;
}

extern "C" {

size_t mfr_emat_workspace_bytes(int B, int maxN, int max_iters)
{
    if (B <= 0 || maxN <= 0) return 0;
    if (max_iters < 1) max_iters = 1;
    return em_ws_layout(B, maxN, max_iters).total;
}

// host: the table of the normalised MAGSAC++ loss and IRLS weight (lut[2 j], lut[2 j + 1]) over u_j = j / M = r^2 / (k sigma_max)^2,
// n = 4 degrees of freedom, k = 3.64 (the 0.99 quantile).  x = u k^2 / 2:
//   loss(u)   = [gamma(5/2, x) + x (Gamma(3/2, x) - Gamma(3/2, k^2/2))] / gamma(5/2, k^2/2) - 1      in [-1, 0]
//   weight(u) = (Gamma(3/2, x) - Gamma(3/2, k^2/2)) / (Gamma(3/2, 0) - Gamma(3/2, k^2/2))            = d loss / d r^2 up to a constant
// with Gamma(1/2, x) = sqrt(pi) erfc(sqrt x), Gamma(a + 1, x) = a Gamma(a, x) + x^a e^-x.  Copy it to the device once and hand it to
// mfr_emat_solve_batch (the kernels only interpolate in it: no transcendental on the device).
int mfr_magsac_lut(double *lut, int M)
{
    if (!lut || M < 2 || M > MAGSAC_MAX_M) return MFR_E_ARG;
    const double sqrt_pi = 1.7724538509055160273;
    const double kq = 3.64;
    const double xk = 0.5 * kq * kq;
    const double gk = 0.5 * sqrt_pi * erfc(sqrt(xk)) + sqrt(xk) * exp(-xk);
    const double norm = 0.75 * sqrt_pi - (1.5 * gk + xk * sqrt(xk) * exp(-xk));
    const double w0 = 0.5 * sqrt_pi - gk;
    for (int j = 0; j <= M; ++j) {
        const double x = xk * (double)j / (double)M;
        const double sx = sqrt(x), ex = exp(-x);
        const double gu15 = 0.5 * sqrt_pi * erfc(sx) + sx * ex;
        const double gl25 = 0.75 * sqrt_pi - (1.5 * gu15 + x * sx * ex);
        lut[2 * j] = (gl25 + x * (gu15 - gk)) / norm - 1.0;
        lut[2 * j + 1] = (gu15 - gk) / w0;
    }
    lut[0] = -1.0; lut[1] = 1.0; lut[2 * M] = 0.0; lut[2 * M + 1] = 0.0;
    return 0;
}

int mfr_emat_solve_batch(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                         const void *K0, const void *K1, int k_dtype, double pix_thr, double confidence, int max_iters,
                         uint64_t seed, const int64_t *pair_ids, int score_method, const double *magsac_lut, int lut_m,
                         double max_thr_ratio, void *workspace, size_t workspace_bytes,
                         double *R, double *t, int32_t *n_inliers, int32_t *status, uint8_t *inlier_mask,
                         int32_t *best_iter, int32_t *iters_run, int32_t *counts_out, double *losses_out, int32_t *lo_runs,
                         void *stream)
{
    if (!pts0 || !pts1 || !n_corr || !K0 || !K1 || !pair_ids || !workspace || !R || !t || !n_inliers || !status ||
        B <= 0 || maxN <= 0 || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    const bool magsac = (score_method == MFR_EMAT_SCORE_MAGSAC);
    if (!magsac && score_method != MFR_EMAT_SCORE_COUNT) return MFR_E_ARG;
    if (magsac && (!magsac_lut || lut_m < 2 || lut_m > MAGSAC_MAX_M || !(max_thr_ratio >= 1.0))) return MFR_E_ARG;
    if (max_iters < 1) max_iters = 1;
    const EmWs w = em_ws_layout(B, maxN, max_iters);
    if (workspace_bytes < w.total) return MFR_E_WORKSPACE;
    char *ws = (char *)workspace;
    hipStream_t s = (hipStream_t)stream;
    double *x0 = (double *)(ws + w.x0), *x1 = (double *)(ws + w.x1), *thr2 = (double *)(ws + w.thr2);
    double *Es = (double *)(ws + w.Es), *losses = (double *)(ws + w.losses);
    int32_t *nsol = (int32_t *)(ws + w.nsol), *counts = (int32_t *)(ws + w.counts), *bestm = (int32_t *)(ws + w.bestm);
    int32_t *idx = (int32_t *)(ws + w.idx);
    uint8_t *rm = (uint8_t *)(ws + w.rm);
    const double ratio2 = max_thr_ratio * max_thr_ratio;
    hipLaunchKernelGGL(emat_prep_kernel, dim3((maxN + 255) / 256, B), dim3(256), 0, s, pts0, pts1, n_corr, maxN, K0, K1, k_dtype,
                       pix_thr, x0, x1, thr2);
    CHECK_LAUNCH();
    const dim3 hgrid((max_iters + 63) / 64, B), sgrid((max_iters + EM_HYP_PER_WG - 1) / EM_HYP_PER_WG, B);
    hipLaunchKernelGGL(emat_hyp_kernel, hgrid, dim3(64), 0, s, x0, x1, n_corr, maxN, max_iters, seed, pair_ids, Es, nsol);
    CHECK_LAUNCH();
    const int total = B * max_iters;
    hipLaunchKernelGGL(emat_roots_kernel, dim3((total + FPR_HYP_PER_WG - 1) / FPR_HYP_PER_WG), dim3(256), 0, s, Es, nsol, total);
    CHECK_LAUNCH();
    if (magsac) {
        hipLaunchKernelGGL(emat_score_kernel<true>, sgrid, dim3(256), 0, s, x0, x1, n_corr, maxN, max_iters, thr2, Es, nsol,
                           magsac_lut, lut_m, ratio2, counts, bestm, losses);
        CHECK_LAUNCH();
        hipLaunchKernelGGL(emat_select_kernel<true>, dim3(B), dim3(EM_SEL_THREADS), 0, s, x0, x1, n_corr, maxN, max_iters, thr2, confidence, Es,
                           counts, bestm, losses, magsac_lut, lut_m, ratio2, idx, rm, R, t, n_inliers, status, inlier_mask,
                           best_iter, iters_run, lo_runs);
    } else {
        hipLaunchKernelGGL(emat_score_kernel<false>, sgrid, dim3(256), 0, s, x0, x1, n_corr, maxN, max_iters, thr2, Es, nsol,
                           (const double *)nullptr, 2, 1.0, counts, bestm, losses);
        CHECK_LAUNCH();
        hipLaunchKernelGGL(emat_select_kernel<false>, dim3(B), dim3(EM_SEL_THREADS), 0, s, x0, x1, n_corr, maxN, max_iters, thr2, confidence, Es,
                           counts, bestm, losses, (const double *)nullptr, 2, 1.0, idx, rm, R, t, n_inliers, status, inlier_mask,
                           best_iter, iters_run, lo_runs);
    }
    CHECK_LAUNCH();
    if (counts_out)
        if (hipMemcpyAsync(counts_out, counts, sizeof(int32_t) * (size_t)B * max_iters, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return MFR_E_LAUNCH;
    if (losses_out && magsac)
        if (hipMemcpyAsync(losses_out, losses, sizeof(double) * (size_t)B * max_iters, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return MFR_E_LAUNCH;
    return 0;
}

}  // extern "C"
