/*
 * oracle/mfr_oracle_icp.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mfr_oracle.h).
 *
 * ProcrustesSolver's optional refinement, lib/models/matching/pose_solver.py:290-315 (PROCRUSTES.REFINE = True; used by
 * config/matching/scannet/ *_icp.yaml): back-project BOTH whole depth maps (valid = depth > 0, :296-300),
 *   o3d.pipelines.registration.registration_icp(pcl_0, pcl_1, MAX_CORR_DIST, init = RANSAC transform,
 *                                               ICPConvergenceCriteria(1e-4, 1e-4, 30))          (:307-315)
 * and inliers = int(fitness * |pcl_1|) (:319).  Open3D 0.17 (environment.yml:16) is not available offline; its published
 * algorithm (point-to-point ICP) is restated:
 *     result = evaluate(T)                       nearest target point of every transformed source point, kept if the
 *     repeat <= max_iteration times:             distance is < max_correspondence_distance; fitness = #corr / |source|,
 *         T <- kabsch(corr) o T                   inlier_rmse = sqrt(sum d^2 / #corr)
 *         backup = result; result = evaluate(T)
 *         stop when |d fitness| < 1e-4 and |d rmse| < 1e-4
 * Substitutions (PARITY UNPINNED against Open3D): the KD-tree nearest-neighbour query is replaced by an EXACT search over
 * the pixel window of the organised target cloud that can contain points within the radius (the bound is derived in
 * csrc/procrustes_icp.hip), ties -> lowest pixel index; the composed transform is applied to the original source points
 * (Open3D transforms the cloud incrementally); Horn's quaternion method instead of Eigen::umeyama.
 * Summation order = the device's: 256-pixel blocks, wave64 butterflies inside, block sums added in block order.
 */
#include "mfr_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* (prototype in mfr_oracle.h) */
void mfr_ref_kabsch_from_moments(const double *s, double *R, double *t);

#define NACC 17

static void rot_apply(const double *R, const double *t, const double *X, double *Y)
{
    Y[0] = ((R[0] * X[0] + R[1] * X[1]) + R[2] * X[2]) + t[0];
    Y[1] = ((R[3] * X[0] + R[4] * X[1]) + R[5] * X[2]) + t[1];
    Y[2] = ((R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]) + t[2];
}

/* nearest target point of Y within radius r (strictly): returns pixel index or -1, *bd2 = squared distance */
static int nearest(const double *Y, const float *depth1, const double *Tc, int H, int W, const mfr_intr *k1, double r, double *bd2)
{
    const double fx = k1->fx, fy = k1->fy, cx = k1->cx, cy = k1->cy;
    const double Z = Y[2];
    int u0 = 0, u1 = W - 1, v0 = 0, v1 = H - 1;
    if (Z > 2.0 * r) {
        const double uS = fx * (Y[0] / Z) + cx, vS = fy * (Y[1] / Z) + cy;
        double au = uS - cx, av = vS - cy;
        if (au < 0.0) au = -au;
        if (av < 0.0) av = -av;
        const double wu = ceil((r * (fx + au)) / (Z - r)) + 1.0, wv = ceil((r * (fy + av)) / (Z - r)) + 1.0;
        double lo = floor(uS) - wu, hi = ceil(uS) + wu;
        if (lo < 0.0) lo = 0.0;
        if (hi > (double)(W - 1)) hi = (double)(W - 1);
        if (!(lo <= hi)) return -1;
        u0 = (int)lo; u1 = (int)hi;
        lo = floor(vS) - wv; hi = ceil(vS) + wv;
        if (lo < 0.0) lo = 0.0;
        if (hi > (double)(H - 1)) hi = (double)(H - 1);
        if (!(lo <= hi)) return -1;
        v0 = (int)lo; v1 = (int)hi;
    } else if (!(Z == Z)) return -1;                              /* NaN pose */
    int best = -1;
    double bd = r * r;
    for (int v = v0; v <= v1; ++v)
        for (int u = u0; u <= u1; ++u) {
            const float zt = depth1[v * W + u];
            if (!(zt > 0.f)) continue;
            double dz = (double)zt - Z;
            if (dz < 0.0) dz = -dz;
            if (!(dz < r)) continue;
            const double *q = Tc + 3 * (size_t)(v * W + u);
            const double d0 = Y[0] - q[0], d1 = Y[1] - q[1], d2 = Y[2] - q[2];
            const double dd = (d0 * d0 + d1 * d1) + d2 * d2;
            if (dd < bd) { bd = dd; best = v * W + u; }
        }
    *bd2 = bd;
    return best;
}

static void evaluate(const float *depth0, const float *depth1, const double *Tc, int H, int W, const void *K0, const void *K1,
                     int k_dtype, const double *R, const double *t, double r, double *tot)
{
    mfr_intr k1;
    mfr_ref_load_intr(K1, k_dtype, &k1);
    const int hw = H * W, nblk = (hw + 255) / 256;
    for (int k = 0; k < NACC; ++k) tot[k] = 0.0;
    for (int blk = 0; blk < nblk; ++blk) {
        double wsum[4][NACC];
        for (int w = 0; w < 4; ++w) {
            double a[64][NACC];
            memset(a, 0, sizeof(a));
            for (int l = 0; l < 64; ++l) {
                const int i = blk * 256 + w * 64 + l;
                if (i >= hw) continue;
                const float d = depth0[i];
                if (!(d > 0.f)) continue;
                int32_t uv[2] = { i % W, i / W };
                double X[3], Y[3], bd2 = 0.0;
                mfr_ref_backproject(uv, &d, 1, K0, k_dtype, X);
                rot_apply(R, t, X, Y);
                const int j = nearest(Y, depth1, Tc, H, W, &k1, r, &bd2);
                if (j < 0) continue;
                const double *q = Tc + 3 * (size_t)j;
                double *c = a[l];
                c[0] = 1.0;
                for (int e = 0; e < 3; ++e) { c[1 + e] = Y[e]; c[4 + e] = q[e]; }
                for (int e = 0; e < 3; ++e) for (int f = 0; f < 3; ++f) c[7 + 3 * e + f] = Y[e] * q[f];
                c[16] = bd2;
            }
            for (int off = 32; off >= 1; off >>= 1) {
                double tmp[64][NACC];
                for (int l = 0; l < 64; ++l) for (int k = 0; k < NACC; ++k) tmp[l][k] = a[l][k] + a[l ^ off][k];
                memcpy(a, tmp, sizeof(tmp));
            }
            for (int k = 0; k < NACC; ++k) wsum[w][k] = a[0][k];
        }
        for (int k = 0; k < NACC; ++k) tot[k] = tot[k] + (((wsum[0][k] + wsum[1][k]) + wsum[2][k]) + wsum[3][k]);
    }
}

/* R, t: in = initial transform (RANSAC result), out = refined.  Returns number of update steps taken. */
int mfr_ref_procrustes_icp(const float *depth0, const float *depth1, int H, int W, const void *K0, const void *K1, int k_dtype,
                           double max_dist, double rel_fitness, double rel_rmse, int max_iter, double R[9], double t[3],
                           int *n_inliers, double *fitness_out, double *rmse_out)
{
    const int hw = H * W;
    double *Tc = (double *)malloc(sizeof(double) * 3 * (size_t)hw);
    int nS = 0, nT = 0;
    for (int i = 0; i < hw; ++i) {
        if (depth0[i] > 0.f) ++nS;
        Tc[3 * i] = Tc[3 * i + 1] = Tc[3 * i + 2] = 0.0;
        if (depth1[i] > 0.f) {
            ++nT;
            int32_t uv[2] = { i % W, i / W };
            mfr_ref_backproject(uv, depth1 + i, 1, K1, k_dtype, Tc + 3 * i);
        }
    }
    double tot[NACC], fit_prev = 0.0, rmse_prev = 0.0, fit = 0.0, rmse = 0.0;
    int k;
    for (k = 0; ; ++k) {
        evaluate(depth0, depth1, Tc, H, W, K0, K1, k_dtype, R, t, max_dist, tot);
        const double n = tot[0];
        fit = (nS > 0) ? n / (double)nS : 0.0;
        rmse = (n > 0.0) ? sqrt(tot[16] / n) : 0.0;
        if (k >= 1) {
            double df = fit_prev - fit, dr = rmse_prev - rmse;
            if (df < 0.0) df = -df;
            if (dr < 0.0) dr = -dr;
            if (df < rel_fitness && dr < rel_rmse) break;
        }
        if (k >= max_iter) break;
        if (n >= 3.0) {
            double U[9], Ut[3], Rn[9], tn[3];
            mfr_ref_kabsch_from_moments(tot, U, Ut);
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) Rn[3 * i + j] = (U[3 * i] * R[j] + U[3 * i + 1] * R[3 + j]) + U[3 * i + 2] * R[6 + j];
                tn[i] = ((U[3 * i] * t[0] + U[3 * i + 1] * t[1]) + U[3 * i + 2] * t[2]) + Ut[i];
            }
            memcpy(R, Rn, sizeof(Rn)); memcpy(t, tn, sizeof(tn));
        }
        fit_prev = fit; rmse_prev = rmse;
    }
    free(Tc);
    *n_inliers = (int)(fit * (double)nT);                          /* :319 int(res.fitness * len(pcl_1.points)) */
    if (fitness_out) *fitness_out = fit;
    if (rmse_out) *rmse_out = rmse;
    return k;
}
