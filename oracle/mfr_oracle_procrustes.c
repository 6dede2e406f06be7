/*
 * oracle/mfr_oracle_procrustes.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mfr_oracle.h).
 *
 * ProcrustesSolver.estimate_pose, lib/models/matching/pose_solver.py:238-320 (REFINE = False, the
 * Map-free setting: config/matching/mapfree/sg_procrustes_dptkitti.yaml):
 *   int-truncate both views (:248-249), depth gather (:256-258), valid = d0 > depth0.min() &
 *   d1 > depth1.min() (:261, quirk Q6), back-project both (:273-274),
 *   o3d registration_ransac_based_on_correspondence(pcl0, pcl1, identity correspondences,
 *   MAX_CORR_DIST, RANSACConvergenceCriteria()) (:276-287), inliers = int(fitness * N) (:288).
 * Open3D 0.17 (environment.yml:16) is not available offline: its published algorithm is restated --
 * ransac_n = 3 point-to-point (Kabsch/Umeyama without scale), fitness = inliers / N, best =
 * higher fitness then lower inlier RMSE, exit at ceil(log(1-0.999) / log(1 - fitness^3)) iterations,
 * final re-fit on the inliers of the best model.  Substitutions: Philox sampling (Open3D's RNG is
 * unseeded / OpenMP-parallel, i.e. not reproducible upstream either), Horn's quaternion method with a
 * fixed-sweep Jacobi eigen-solver instead of Eigen's JacobiSVD, iteration cap max_iters (Open3D:
 * 100000).  PARITY UNPINNED against Open3D; pinned by known-answer geometry.
 */
#include "mfr_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

void mfr_ref_sample_distinct(uint64_t seed, uint64_t pair_id, uint32_t iter, int n, int k, int *out);
double mfr_ref_det_log(double x);
/* (prototype in mfr_oracle.h) */
float mfr_ref_depth_min(const float *depth, int hw);

/* symmetric 4x4 eigen-decomposition, cyclic Jacobi, fixed 10 sweeps; returns unit eigenvector of
 * the largest eigenvalue */
static void jacobi4_maxvec(double A[4][4], double q[4])
{
    double V[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 10; ++sweep)
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                double apq = A[p][r];
                if (apq == 0.0) continue;
                double theta = (A[r][r] - A[p][p]) / (2.0 * apq);
                double at = theta < 0.0 ? -theta : theta;
                double t = 1.0 / (at + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {            /* A <- A J */
                    double akp = A[k][p], akq = A[k][r];
                    A[k][p] = c * akp - s * akq; A[k][r] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; ++k) {            /* A <- J^T A */
                    double apk = A[p][k], aqk = A[r][k];
                    A[p][k] = c * apk - s * aqk; A[r][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; ++k) {
                    double vkp = V[k][p], vkq = V[k][r];
                    V[k][p] = c * vkp - s * vkq; V[k][r] = s * vkp + c * vkq;
                }
            }
    int b = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[b][b]) b = i;
    double nn = sqrt(((V[0][b] * V[0][b] + V[1][b] * V[1][b]) + V[2][b] * V[2][b]) + V[3][b] * V[3][b]);
    for (int i = 0; i < 4; ++i) q[i] = V[i][b] / nn;
}

/* Horn 1987: rigid transform q = R p + t minimising sum |R p_i + t - q_i|^2 from the 15 moments
 * s = {n, sum p (3), sum q (3), sum p_a q_b (9)} */
static void kabsch_from_moments(const double *s, double *R, double *t)
{
    double n = s[0], pc[3] = { s[1] / n, s[2] / n, s[3] / n }, qc[3] = { s[4] / n, s[5] / n, s[6] / n };
    double S[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] = s[7 + 3 * a + b] - n * pc[a] * qc[b];
    double N[4][4];
    N[0][0] = (S[0][0] + S[1][1]) + S[2][2];
    N[0][1] = S[1][2] - S[2][1]; N[0][2] = S[2][0] - S[0][2]; N[0][3] = S[0][1] - S[1][0];
    N[1][1] = (S[0][0] - S[1][1]) - S[2][2]; N[1][2] = S[0][1] + S[1][0]; N[1][3] = S[2][0] + S[0][2];
    N[2][2] = (-S[0][0] + S[1][1]) - S[2][2]; N[2][3] = S[1][2] + S[2][1];
    N[3][3] = (-S[0][0] - S[1][1]) + S[2][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < i; ++j) N[i][j] = N[j][i];
    double q[4];
    jacobi4_maxvec(N, q);
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i) t[i] = qc[i] - ((R[3 * i] * pc[0] + R[3 * i + 1] * pc[1]) + R[3 * i + 2] * pc[2]);
}

void mfr_ref_kabsch_from_moments(const double *s, double *R, double *t) { kabsch_from_moments(s, R, t); }

static inline double dist2(const double *R, const double *t, const double *p, const double *q)
{
    double d0 = (((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0]) - q[0];
    double d1 = (((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1]) - q[1];
    double d2 = (((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2]) - q[2];
    return (d0 * d0 + d1 * d1) + d2 * d2;
}

static void wave_finish(double a[64][16], int nacc, double *out)
{
    for (int off = 32; off >= 1; off >>= 1) {
        double tmp[64][16];
        for (int l = 0; l < 64; ++l) for (int k = 0; k < nacc; ++k) tmp[l][k] = a[l][k] + a[l ^ off][k];
        memcpy(a, tmp, sizeof(tmp));
    }
    for (int k = 0; k < nacc; ++k) out[k] = a[0][k];
}

/* inlier count and squared-error sum of a model.  Summation order = the device's: points are
 * processed in tiles of 512 (the LDS tile), wave64 order inside a tile, tile sums added in order. */
#define PR_TILE 512
static int evaluate(const double *P, const double *Q, int n, const double *R, const double *t, double thr2, double *err2)
{
    int cnt = 0;
    double total = 0.0;
    for (int base = 0; base < n; base += PR_TILE) {
        double a[64][16]; memset(a, 0, sizeof(a));
        int tn = (n - base < PR_TILE) ? n - base : PR_TILE;
        for (int i = 0; i < tn; ++i) {
            double d = dist2(R, t, P + 3 * (base + i), Q + 3 * (base + i));
            if (d < thr2) { ++cnt; a[i & 63][0] = a[i & 63][0] + d; }
        }
        double ts; wave_finish(a, 1, &ts);
        total = total + ts;
    }
    *err2 = total;
    return cnt;
}

static void moments3(const double *P, const double *Q, const int *s, double *m)
{
    for (int k = 0; k < 16; ++k) m[k] = 0.0;
    for (int j = 0; j < 3; ++j) {
        const double *p = P + 3 * s[j], *q = Q + 3 * s[j];
        m[0] = m[0] + 1.0;
        for (int a = 0; a < 3; ++a) { m[1 + a] = m[1 + a] + p[a]; m[4 + a] = m[4 + a] + q[a]; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) m[7 + 3 * a + b] = m[7 + 3 * a + b] + p[a] * q[b];
    }
}

/* lift: both views int-truncated, validity vs each map's minimum (Q6), ordered compaction */
int mfr_ref_procrustes_lift(const float *pts0, const float *pts1, int n, const float *depth0, const float *depth1, int H, int W,
                            const void *K0, const void *K1, int k_dtype, double *P, double *Q)
{
    float m0 = mfr_ref_depth_min(depth0, H * W), m1 = mfr_ref_depth_min(depth1, H * W);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        int32_t u0 = (int32_t)pts0[2 * i], v0 = (int32_t)pts0[2 * i + 1], u1 = (int32_t)pts1[2 * i], v1 = (int32_t)pts1[2 * i + 1];
        if (u0 < 0 || u0 >= W || v0 < 0 || v0 >= H || u1 < 0 || u1 >= W || v1 < 0 || v1 >= H) continue;
        float d0 = depth0[v0 * W + u0], d1 = depth1[v1 * W + u1];
        if (!(d0 > m0) || !(d1 > m1)) continue;                              /* :261 */
        int32_t a[2] = { u0, v0 }, b[2] = { u1, v1 };
        if (mfr_ref_backproject(a, &d0, 1, K0, k_dtype, P + 3 * m)) return -1;
        if (mfr_ref_backproject(b, &d1, 1, K1, k_dtype, Q + 3 * m)) return -1;
        ++m;
    }
    return m;
}

int mfr_ref_procrustes_ransac(const double *P, const double *Q, int n, double max_dist, double conf, int max_iters,
                              uint64_t seed, uint64_t pair_id, double R[9], double t[3], int *n_inl,
                              int *best_iter, int *iters_run, int32_t *counts)
{
    const double thr2 = max_dist * max_dist;
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;            /* Open3D default: identity, fitness 0 */
    for (int i = 0; i < 3; ++i) t[i] = 0.0;
    *n_inl = 0; if (best_iter) *best_iter = -1; if (iters_run) *iters_run = 0;
    if (n < 3) return MFR_ST_TOO_FEW;
    if (max_iters < 1) max_iters = 1;
    double bR[9], bt[3], berr = 0.0; int bcnt = 0, bit = -1;
    int est_k = max_iters, it;
    const double lnum = mfr_ref_det_log(1.0 - conf);
    for (it = 0; it < est_k; ++it) {
        int s[3];
        if (n == 3) { s[0] = 0; s[1] = 1; s[2] = 2; } else mfr_ref_sample_distinct(seed, pair_id, (uint32_t)it, n, 3, s);
        double m[16], hR[9], ht[3], e2;
        moments3(P, Q, s, m);
        kabsch_from_moments(m, hR, ht);
        int cnt = evaluate(P, Q, n, hR, ht, thr2, &e2);
        if (counts) counts[it] = cnt;
        /* IsBetterRANSACThan: fitness higher, or equal fitness and lower inlier rmse (rmse^2 = e2/cnt) */
        int better = 0;
        if (cnt > bcnt) better = 1;
        else if (cnt == bcnt && cnt > 0 && e2 < berr) better = 1;
        if (better) {
            bcnt = cnt; berr = e2; bit = it; memcpy(bR, hR, 72); memcpy(bt, ht, 24);
            double ratio = (double)cnt / (double)n, r3 = (ratio * ratio) * ratio;
            int k;
            if (r3 >= 1.0) k = 0;
            else {
                double kd = lnum / mfr_ref_det_log(1.0 - r3);
                k = (kd < (double)est_k) ? (int)ceil(kd) : est_k;
            }
            if (k < est_k) est_k = k;
        }
    }
    if (counts) for (int k = it; k < max_iters; ++k) counts[k] = -1;
    if (iters_run) *iters_run = it;
    if (best_iter) *best_iter = bit;
    if (bit < 0) return MFR_ST_OK;                                           /* identity, 0 inliers (Open3D behaviour) */
    /* final re-fit on the inliers of the best model, then re-evaluate */
    {
        double a[64][16]; memset(a, 0, sizeof(a));
        int q = 0;
        for (int i = 0; i < n; ++i) {
            if (!(dist2(bR, bt, P + 3 * i, Q + 3 * i) < thr2)) continue;
            double *acc = a[q & 63]; ++q;
            const double *p = P + 3 * i, *qq = Q + 3 * i;
            acc[0] = acc[0] + 1.0;
            for (int c = 0; c < 3; ++c) { acc[1 + c] = acc[1 + c] + p[c]; acc[4 + c] = acc[4 + c] + qq[c]; }
            for (int c = 0; c < 3; ++c) for (int d = 0; d < 3; ++d) acc[7 + 3 * c + d] = acc[7 + 3 * c + d] + p[c] * qq[d];
        }
        double m[16]; wave_finish(a, 16, m);
        if (q >= 3) kabsch_from_moments(m, bR, bt);
    }
    double e2;
    int cnt = evaluate(P, Q, n, bR, bt, thr2, &e2);
    memcpy(R, bR, 72); memcpy(t, bt, 24);
    *n_inl = cnt;                                                            /* int(fitness * N) (:288) */
    return MFR_ST_OK;
}

int mfr_ref_procrustes_solve(const float *pts0, const float *pts1, int n, const float *depth0, const float *depth1, int H, int W,
                             const void *K0, const void *K1, int k_dtype, double max_dist, double conf, int max_iters,
                             uint64_t seed, uint64_t pair_id, double R[9], double t[3], int *n_inl)
{
    for (int i = 0; i < 9; ++i) R[i] = NAN;
    for (int i = 0; i < 3; ++i) t[i] = NAN;
    *n_inl = 0;
    if (n < 3) return MFR_ST_TOO_FEW;                                        /* :252-253 */
    double *P = (double *)malloc(sizeof(double) * 3 * (size_t)n), *Q = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    int m = mfr_ref_procrustes_lift(pts0, pts1, n, depth0, depth1, H, W, K0, K1, k_dtype, P, Q);
    int st;
    if (m < 3) st = MFR_ST_BAD_DEPTH;                                        /* :262-263 */
    else st = mfr_ref_procrustes_ransac(P, Q, m, max_dist, conf, max_iters, seed, pair_id, R, t, n_inl, NULL, NULL, NULL);
    free(P); free(Q);
    return st;
}
