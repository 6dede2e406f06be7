/*
 * oracle/mfr_oracle_emat.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mfr_oracle.h).
 *
 * Essential-matrix leg of the reference: EssentialMatrixSolver.estimate_pose,
 * lib/models/matching/pose_solver.py:29-61 =
 *     K-normalise (:39-40), thr = PIX_THRESHOLD / mean(fx0,fy1,fy0,fx1) (:43, quirk Q8),
 *     cv.findEssentialMat(USAC_MAGSAC, prob) (:46-48), cv.recoverPose per E (:56-60).
 * The two cv calls live in opencv-python==4.8.0.74 (not available offline): restated from the
 * published algorithms -- Nister's 5-point solver, RANSAC with OpenCV's adaptive iteration cap,
 * Sampson distances, Horn's closed-form E decomposition, cheirality vote.
 *
 * Model quality and local optimisation (score_method):
 *   MFR_EMAT_SCORE_MAGSAC (0, what the reference asks OpenCV for: method=cv.USAC_MAGSAC, pose_solver.py:46-48)
 *       MAGSAC++ (Barath, Noskova, Ivashechkin, Matas, CVPR 2020): every hypothesis is scored by the sigma-marginalised loss
 *       (noise scale integrated over (0, sigma_max], chi distribution with 4 degrees of freedom, k = 3.64 = its 0.99 quantile)
 *       through a lookup table of the incomplete gamma functions -- OpenCV's USAC does the same (its GammaValues table) --, the
 *       model with the smallest total loss wins, the number of points under the caller's threshold ("tentative inliers")
 *       drives the adaptive iteration cap, and sigma-consensus++ (iteratively re-weighted least squares with the MAGSAC++
 *       weights = d loss / d r^2) is the local optimisation: run on every new best model from iteration MFR_MAGSAC_LO_START
 *       on (USAC: max_iters_before_LO = 100) and once more at the end if the winner never went through it.  The least-squares
 *       problem inside each re-weighting round is solved on the essential manifold (one damped Gauss-Newton step on (R, unit t)
 *       per round, accepted when the MAGSAC++ loss decreases) instead of USAC's weighted linear solver + projection.
 *       k * sigma_max = max_thr_ratio * threshold; the final mask is residual^2 < threshold^2 (USAC's strict compare).
 *   MFR_EMAT_SCORE_COUNT (1, rounds 1-3; kept for A/B): inlier count at the same threshold + one LM polish of (R,t) on the
 *       inliers, kept if it does not lose inliers.
 * opencv-python 4.8.0.74 is not available offline, so USAC's exact constants (how `threshold` maps to its maximum sigma, its
 * table quantisation, LO sample size 50, polisher) are NOT reproduced: PARITY UNPINNED against OpenCV; pinned by known-answer
 * synthetic geometry (tests/test_oracle_known_answers.py); tests/external/gen_cv_golden.py dumps OpenCV's E / mask / inlier
 * count for the same seeded sets wherever cv2 exists.
 */
#include "mfr_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

int mfr_ref_update_num_iters(double p, double ep, int model_points, int max_iters);
void mfr_ref_sample_distinct(uint64_t seed, uint64_t pair_id, uint32_t iter, int n, int k, int *out);
int mfr_ref_poly_real_roots(const double *c, int deg, double *roots);

/* monomial bookkeeping (generated; degree<=1: [x,y,z,1]; degree<=2: [xx,xy,xz,yy,yz,zz,x,y,z,1];
 * degree<=3: [xxx,xxy,xxz,xyy,xyz,xzz,yyy,yyz,yzz,zzz,xx,xy,xz,yy,yz,zz,x,y,z,1]) */
static const int IDX11[4][4] = { {0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9} };
static const int IDX21[10][4] = { {0, 1, 2, 10}, {1, 3, 4, 11}, {2, 4, 5, 12}, {3, 6, 7, 13}, {4, 7, 8, 14},
                                  {5, 8, 9, 15}, {10, 11, 12, 16}, {11, 13, 14, 17}, {12, 14, 15, 18}, {16, 17, 18, 19} };
/* Nister's column order [x3,y3,x2y,xy2,x2z,x2,y2z,y2,xyz,xy, xz2,xz,x,yz2,yz,y,z3,z2,z,1] -> degree-3 index */
static const int NPERM[20] = { 0, 6, 1, 3, 2, 10, 7, 13, 4, 11, 5, 12, 16, 8, 14, 17, 9, 15, 18, 19 };

static void p_mul11(const double *a, const double *b, double *o)      /* o[10] += a[4]*b[4] */
{
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[IDX11[i][j]] = o[IDX11[i][j]] + a[i] * b[j];
}
static void p_mul21(const double *a, const double *b, double *o)      /* o[20] += a[10]*b[4] */
{
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) o[IDX21[i][j]] = o[IDX21[i][j]] + a[i] * b[j];
}

/* Nister (PAMI 2004) 5-point relative pose.  x0, x1: 5 normalised points each (x,y).
 * Returns up to 10 essential matrices (row-major), x1^T E x0 = 0. */
int mfr_ref_fivept(const double *x0, const double *x1, double *Es)
{
    /* 5 x 9 epipolar constraints, rows q = [x1x0, x1y0, x1, y1x0, y1y0, y1, x0, y0, 1] */
    double A[5][9];
    for (int i = 0; i < 5; ++i) {
        double a = x0[2 * i], b = x0[2 * i + 1], c = x1[2 * i], d = x1[2 * i + 1];
        A[i][0] = c * a; A[i][1] = c * b; A[i][2] = c; A[i][3] = d * a; A[i][4] = d * b; A[i][5] = d;
        A[i][6] = a; A[i][7] = b; A[i][8] = 1.0;
    }
    /* null space by Gauss-Jordan with full pivoting */
    int colp[9]; for (int j = 0; j < 9; ++j) colp[j] = j;
    for (int r = 0; r < 5; ++r) {
        int pr = r, pc = r; double best = -1.0;
        for (int i = r; i < 5; ++i) for (int j = r; j < 9; ++j) {
            double v = A[i][j] < 0.0 ? -A[i][j] : A[i][j];
            if (v > best) { best = v; pr = i; pc = j; }
        }
        if (!(best > 1e-300)) return 0;
        if (pr != r) for (int j = 0; j < 9; ++j) { double tmp = A[r][j]; A[r][j] = A[pr][j]; A[pr][j] = tmp; }
        if (pc != r) { for (int i = 0; i < 5; ++i) { double tmp = A[i][r]; A[i][r] = A[i][pc]; A[i][pc] = tmp; }
                       int ti = colp[r]; colp[r] = colp[pc]; colp[pc] = ti; }
        double inv = 1.0 / A[r][r];
        for (int j = 0; j < 9; ++j) A[r][j] = A[r][j] * inv;
        for (int i = 0; i < 5; ++i) if (i != r) {
            double f = A[i][r];
            for (int j = 0; j < 9; ++j) A[i][j] = A[i][j] - f * A[r][j];
        }
    }
    double N[4][9];                                  /* basis X,Y,Z,W of the null space */
    for (int k = 0; k < 4; ++k) {
        double v[9];
        for (int j = 0; j < 9; ++j) v[j] = 0.0;
        v[5 + k] = 1.0;
        for (int r = 0; r < 5; ++r) v[r] = -A[r][5 + k];
        for (int j = 0; j < 9; ++j) N[k][colp[j]] = v[j];
    }
    /* E entries as degree-1 polynomials in (x,y,z): E = xX + yY + zZ + W */
    double Ep[9][4];
    for (int e = 0; e < 9; ++e) for (int k = 0; k < 4; ++k) Ep[e][k] = N[k][e];
    /* 10 cubic constraints in the degree-3 monomial order */
    double C[10][20];
    memset(C, 0, sizeof(C));
    {   /* det(E) */
        double m[10], neg[4];
        #define MINOR(a, b, c, d) do { memset(m, 0, sizeof(m)); p_mul11(Ep[a], Ep[b], m); \
            for (int q = 0; q < 4; ++q) neg[q] = -Ep[c][q]; p_mul11(neg, Ep[d], m); } while (0)
        MINOR(4, 8, 5, 7); p_mul21(m, Ep[0], C[0]);
        MINOR(5, 6, 3, 8); p_mul21(m, Ep[1], C[0]);
        MINOR(3, 7, 4, 6); p_mul21(m, Ep[2], C[0]);
        #undef MINOR
    }
    {   /* (E E^T - 1/2 trace(E E^T) I) E = 0 */
        double EEt[3][3][10];
        memset(EEt, 0, sizeof(EEt));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k)
            p_mul11(Ep[3 * i + k], Ep[3 * j + k], EEt[i][j]);
        double tr[10];
        for (int q = 0; q < 10; ++q) tr[q] = (EEt[0][0][q] + EEt[1][1][q]) + EEt[2][2][q];
        for (int i = 0; i < 3; ++i) for (int q = 0; q < 10; ++q) EEt[i][i][q] = EEt[i][i][q] - 0.5 * tr[q];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k)
            p_mul21(EEt[i][k], Ep[3 * k + j], C[1 + 3 * i + j]);
    }
    /* to Nister's column order, Gauss-Jordan on the first 10 columns (partial pivoting) */
    double M[10][20];
    for (int r = 0; r < 10; ++r) for (int c = 0; c < 20; ++c) M[r][c] = C[r][NPERM[c]];
    for (int c = 0; c < 10; ++c) {
        int pr = c; double best = -1.0;
        for (int i = c; i < 10; ++i) { double v = M[i][c] < 0.0 ? -M[i][c] : M[i][c]; if (v > best) { best = v; pr = i; } }
        if (!(best > 1e-300)) return 0;
        if (pr != c) for (int j = 0; j < 20; ++j) { double tmp = M[c][j]; M[c][j] = M[pr][j]; M[pr][j] = tmp; }
        double inv = 1.0 / M[c][c];
        for (int j = 0; j < 20; ++j) M[c][j] = M[c][j] * inv;
        for (int i = 0; i < 10; ++i) if (i != c) {
            double f = M[i][c];
            for (int j = 0; j < 20; ++j) M[i][j] = M[i][j] - f * M[c][j];
        }
    }
    /* B(z) rows <k> = e - z f, <l> = g - z h, <m> = i - z j ; entries ascending in z */
    double Bx[3][4], By[3][4], B1[3][5];
    const int hi[3] = { 4, 6, 8 }, lo[3] = { 5, 7, 9 };
    for (int r = 0; r < 3; ++r) {
        const double *e = M[hi[r]], *f = M[lo[r]];
        Bx[r][0] = e[12]; Bx[r][1] = e[11] - f[12]; Bx[r][2] = e[10] - f[11]; Bx[r][3] = -f[10];
        By[r][0] = e[15]; By[r][1] = e[14] - f[15]; By[r][2] = e[13] - f[14]; By[r][3] = -f[13];
        B1[r][0] = e[19]; B1[r][1] = e[18] - f[19]; B1[r][2] = e[17] - f[18]; B1[r][3] = e[16] - f[17]; B1[r][4] = -f[16];
    }
    /* det B(z): degree 10 */
    double P[11];
    for (int q = 0; q < 11; ++q) P[q] = 0.0;
    {
        /* cofactor polynomials: c0 = By1*B12 - B11*By2 (deg 7), c1 = Bx1*B12 - B11*Bx2 (deg 7), c2 = Bx1*By2 - By1*Bx2 (deg 6) */
        double c0[8], c1[8], c2[7];
        for (int q = 0; q < 8; ++q) { c0[q] = 0.0; c1[q] = 0.0; }
        for (int q = 0; q < 7; ++q) c2[q] = 0.0;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 5; ++b) {
            c0[a + b] = c0[a + b] + (By[1][a] * B1[2][b] - B1[1][b] * By[2][a]);
            c1[a + b] = c1[a + b] + (Bx[1][a] * B1[2][b] - B1[1][b] * Bx[2][a]);
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b)
            c2[a + b] = c2[a + b] + (Bx[1][a] * By[2][b] - By[1][a] * Bx[2][b]);
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 8; ++b) P[a + b] = P[a + b] + (Bx[0][a] * c0[b] - By[0][a] * c1[b]);
        for (int a = 0; a < 5; ++a) for (int b = 0; b < 7; ++b) P[a + b] = P[a + b] + B1[0][a] * c2[b];
    }
    double roots[10];
    int nr = mfr_ref_poly_real_roots(P, 10, roots);
    int ns = 0;
    for (int r = 0; r < nr; ++r) {
        double z = roots[r];
        double bx[3], by[3], b1[3];
        for (int k = 0; k < 3; ++k) {
            bx[k] = ((Bx[k][3] * z + Bx[k][2]) * z + Bx[k][1]) * z + Bx[k][0];
            by[k] = ((By[k][3] * z + By[k][2]) * z + By[k][1]) * z + By[k][0];
            b1[k] = (((B1[k][4] * z + B1[k][3]) * z + B1[k][2]) * z + B1[k][1]) * z + B1[k][0];
        }
        /* null vector of B(z): the best-conditioned cross product of two rows */
        double v[3] = { 0, 0, 0 }, bestw = -1.0;
        for (int a = 0; a < 3; ++a) {
            int p = a, q = (a + 1) % 3;
            double w0 = by[p] * b1[q] - b1[p] * by[q];
            double w1 = b1[p] * bx[q] - bx[p] * b1[q];
            double w2 = bx[p] * by[q] - by[p] * bx[q];
            double aw = w2 < 0.0 ? -w2 : w2;
            if (aw > bestw) { bestw = aw; v[0] = w0; v[1] = w1; v[2] = w2; }
        }
        if (!(bestw > 0.0)) continue;
        double x = v[0] / v[2], y = v[1] / v[2];
        double *E = Es + 9 * ns, nn = 0.0;
        for (int e = 0; e < 9; ++e) {
            E[e] = ((x * Ep[e][0] + y * Ep[e][1]) + z * Ep[e][2]) + Ep[e][3];
            nn = nn + E[e] * E[e];
        }
        if (!(nn > 0.0) || !(nn < 1e300)) continue;
        double s = 1.0 / sqrt(nn);
        for (int e = 0; e < 9; ++e) E[e] = E[e] * s;
        ++ns;
    }
    return ns;
}

/* squared Sampson distance of (x0, x1) to E (x1^T E x0 = 0), normalised coordinates */
static inline double sampson2(const double *E, double a, double b, double c, double d)
{
    double Ex0 = (E[0] * a + E[1] * b) + E[2], Ex1 = (E[3] * a + E[4] * b) + E[5], Ex2 = (E[6] * a + E[7] * b) + E[8];
    double Et0 = (E[0] * c + E[3] * d) + E[6], Et1 = (E[1] * c + E[4] * d) + E[7];
    double num = (c * Ex0 + d * Ex1) + Ex2;
    double den = ((Ex0 * Ex0 + Ex1 * Ex1) + Et0 * Et0) + Et1 * Et1;
    return (num * num) / den;
}

static void skew_mul(const double *t, const double *R, double *E)   /* E = [t]x R */
{
    for (int j = 0; j < 3; ++j) {
        E[j]     = t[1] * R[6 + j] - t[2] * R[3 + j];
        E[3 + j] = t[2] * R[j]     - t[0] * R[6 + j];
        E[6 + j] = t[0] * R[3 + j] - t[1] * R[j];
    }
}

/* Horn 1990: E ~ [b]x R.  b from b b^T = 1/2 tr(E E^T) I - E E^T, R = (cof(E) - [b]x E) / (b.b) for +-b. */
int mfr_ref_emat_decompose(const double *E, double *Ra, double *Rb, double *tu)
{
    double EEt[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        EEt[3 * i + j] = (E[3 * i] * E[3 * j] + E[3 * i + 1] * E[3 * j + 1]) + E[3 * i + 2] * E[3 * j + 2];
    double htr = 0.5 * ((EEt[0] + EEt[4]) + EEt[8]);
    double bb[9];
    for (int i = 0; i < 9; ++i) bb[i] = -EEt[i];
    bb[0] = bb[0] + htr; bb[4] = bb[4] + htr; bb[8] = bb[8] + htr;
    int k = 0;
    if (bb[4] > bb[0]) k = 1;
    if (bb[8] > bb[4 * k]) k = 2;
    if (!(bb[4 * k] > 0.0)) return -1;
    double s = sqrt(bb[4 * k]);
    double b[3] = { bb[k] / s, bb[3 + k] / s, bb[6 + k] / s };
    double C[9];
    C[0] = E[4] * E[8] - E[5] * E[7]; C[1] = -(E[3] * E[8] - E[5] * E[6]); C[2] = E[3] * E[7] - E[4] * E[6];
    C[3] = -(E[1] * E[8] - E[2] * E[7]); C[4] = E[0] * E[8] - E[2] * E[6]; C[5] = -(E[0] * E[7] - E[1] * E[6]);
    C[6] = E[1] * E[5] - E[2] * E[4]; C[7] = -(E[0] * E[5] - E[2] * E[3]); C[8] = E[0] * E[4] - E[1] * E[3];
    double bE[9];
    skew_mul(b, E, bE);
    double b2 = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    for (int i = 0; i < 9; ++i) { Ra[i] = (C[i] - bE[i]) / b2; Rb[i] = (C[i] + bE[i]) / b2; }
    double nb = sqrt(b2);
    tu[0] = b[0] / nb; tu[1] = b[1] / nb; tu[2] = b[2] / nb;
    return 0;
}

/* depths of the closest points of the two rays: min | l0 (R x0h) + t - l1 x1h |  (cheirality of
 * cv::recoverPose, restated with a 2x2 closed form instead of DLT triangulation) */
static inline int cheirality(const double *R, const double *t, double a, double b, double c, double d)
{
    double p[3] = { (R[0] * a + R[1] * b) + R[2], (R[3] * a + R[4] * b) + R[5], (R[6] * a + R[7] * b) + R[8] };
    double q[3] = { c, d, 1.0 };
    double pp = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2], qq = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    double pq = (p[0] * q[0] + p[1] * q[1]) + p[2] * q[2];
    double pt = (p[0] * t[0] + p[1] * t[1]) + p[2] * t[2], qt = (q[0] * t[0] + q[1] * t[1]) + q[2] * t[2];
    double det = pp * qq - pq * pq;
    if (!(det > 1e-18 * pp * qq)) return 0;
    double l0 = (pq * qt - qq * pt) / det;        /* depth in camera 0 (along x0h, z = 1) */
    double l1 = (pp * qt - pq * pt) / det;        /* depth in camera 1 */
    return (l0 > 0.0) && (l1 > 0.0);
}

#define NACC 28
/* Sums over points in the order the GPU workgroup forms them (csrc/emat.hip emat_select_kernel, EM_SEL_WAVES = 4 wavefronts of 64 lanes): point i
 * goes to virtual lane i mod 256; the 64 lanes of a wavefront merge in an xor butterfly, the 4 wavefront totals are added in sequence.
 * (Rounds 1-4: one wavefront, lane = i mod 64.) */
#define EM_SEL_WAVES 4
#define EM_SEL_LANES (64 * EM_SEL_WAVES)
typedef struct { double a[EM_SEL_LANES][NACC]; } wacc_t;
static void wacc_finish(wacc_t *w, int nacc, double *out)
{
    for (int g = 0; g < EM_SEL_WAVES; ++g) {
        double (*a)[NACC] = w->a + 64 * g;
        for (int off = 32; off >= 1; off >>= 1) {
            double tmp[64][NACC];
            for (int l = 0; l < 64; ++l) for (int k = 0; k < nacc; ++k) tmp[l][k] = a[l][k] + a[l ^ off][k];
            memcpy(a, tmp, sizeof(tmp));
        }
    }
    for (int k = 0; k < nacc; ++k) {
        double s = w->a[0][k];
        for (int g = 1; g < EM_SEL_WAVES; ++g) s = s + w->a[64 * g][k];
        out[k] = s;
    }
}

static void quat_right(const double *R, const double *dw, double *Rn)
{
    double hx = 0.5 * dw[0], hy = 0.5 * dw[1], hz = 0.5 * dw[2];
    double nn = sqrt(((hx * hx + hy * hy) + hz * hz) + 1.0);
    double w = 1.0 / nn, x = hx / nn, y = hy / nn, z = hz / nn;
    double Q[9];
    Q[0] = 1.0 - 2.0 * (y * y + z * z); Q[1] = 2.0 * (x * y - w * z);       Q[2] = 2.0 * (x * z + w * y);
    Q[3] = 2.0 * (x * y + w * z);       Q[4] = 1.0 - 2.0 * (x * x + z * z); Q[5] = 2.0 * (y * z - w * x);
    Q[6] = 2.0 * (x * z - w * y);       Q[7] = 2.0 * (y * z + w * x);       Q[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
        Rn[3 * i + j] = (R[3 * i] * Q[j] + R[3 * i + 1] * Q[3 + j]) + R[3 * i + 2] * Q[6 + j];
}

static int chol6(const double *A, const double *b, double *x)
{
    double L[36]; memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) {
        double s = A[6 * i + j];
        for (int k = 0; k < j; ++k) s = s - L[6 * i + k] * L[6 * j + k];
        if (i == j) { if (!(s > 0.0)) return -1; L[6 * i + i] = sqrt(s); }
        else L[6 * i + j] = s / L[6 * j + j];
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s = s - L[6 * i + k] * y[k]; y[i] = s / L[6 * i + i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s = s - L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
    return 0;
}

static double emat_cost(const double *x0, const double *x1, const int32_t *idx, int n, const double *R, const double *t)
{
    double E[9]; skew_mul(t, R, E);
    wacc_t *w = (wacc_t *)calloc(1, sizeof(wacc_t));
    for (int i = 0; i < n; ++i) {
        int j = idx[i];
        w->a[i % EM_SEL_LANES][0] = w->a[i % EM_SEL_LANES][0] + sampson2(E, x0[2 * j], x0[2 * j + 1], x1[2 * j], x1[2 * j + 1]);
    }
    double c; wacc_finish(w, 1, &c); free(w);
    return c;
}

/* LM polish of (R, unit t) on the Sampson cost over idx (Gauss-Newton on the epipolar residual
 * with the Sampson denominator frozen per iteration); t is re-normalised after every step. */
int mfr_ref_emat_refine(const double *x0, const double *x1, const int32_t *idx, int n, int max_iter, double *R, double *t)
{
    double lambda = 1e-3;
    double cost = emat_cost(x0, x1, idx, n, R, t);
    if (!(cost == cost)) return -1;
    wacc_t *w = (wacc_t *)malloc(sizeof(wacc_t));
    for (int it = 0; it < max_iter; ++it) {
        double E[9]; skew_mul(t, R, E);
        memset(w, 0, sizeof(*w));
        for (int i = 0; i < n; ++i) {
            int j = idx[i];
            double a = x0[2 * j], b = x0[2 * j + 1], c = x1[2 * j], d = x1[2 * j + 1];
            double Ex0 = (E[0] * a + E[1] * b) + E[2], Ex1 = (E[3] * a + E[4] * b) + E[5], Ex2 = (E[6] * a + E[7] * b) + E[8];
            double Et0 = (E[0] * c + E[3] * d) + E[6], Et1 = (E[1] * c + E[4] * d) + E[7];
            double num = (c * Ex0 + d * Ex1) + Ex2;
            double den = ((Ex0 * Ex0 + Ex1 * Ex1) + Et0 * Et0) + Et1 * Et1;
            double wgt = 1.0 / sqrt(den);
            /* d num / d dw = x0h x (R^T [t]x^T x1h) ; d num / d dt = (R x0h) x x1h */
            double q[3] = { c, d, 1.0 }, p[3] = { a, b, 1.0 };
            double txq[3] = { t[1] * q[2] - t[2] * q[1], t[2] * q[0] - t[0] * q[2], t[0] * q[1] - t[1] * q[0] };
            double u[3] = { -((R[0] * txq[0] + R[3] * txq[1]) + R[6] * txq[2]),
                            -((R[1] * txq[0] + R[4] * txq[1]) + R[7] * txq[2]),
                            -((R[2] * txq[0] + R[5] * txq[1]) + R[8] * txq[2]) };     /* R^T [t]x^T q = -R^T (t x q) */
            double Rp[3] = { (R[0] * p[0] + R[1] * p[1]) + R[2] * p[2], (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2],
                             (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2] };
            double J[6];
            J[0] = (p[1] * u[2] - p[2] * u[1]) * wgt; J[1] = (p[2] * u[0] - p[0] * u[2]) * wgt; J[2] = (p[0] * u[1] - p[1] * u[0]) * wgt;
            J[3] = (Rp[1] * q[2] - Rp[2] * q[1]) * wgt; J[4] = (Rp[2] * q[0] - Rp[0] * q[2]) * wgt; J[5] = (Rp[0] * q[1] - Rp[1] * q[0]) * wgt;
            double r = num * wgt;
            double *acc = w->a[i % EM_SEL_LANES];
            int qq = 0;
            for (int rr = 0; rr < 6; ++rr) for (int cc = rr; cc < 6; ++cc, ++qq) acc[qq] = acc[qq] + J[rr] * J[cc];
            for (int rr = 0; rr < 6; ++rr, ++qq) acc[qq] = acc[qq] + J[rr] * r;
        }
        double s[27]; wacc_finish(w, 27, s);
        double H[36], g[6];
        { int qq = 0;
          for (int rr = 0; rr < 6; ++rr) for (int cc = rr; cc < 6; ++cc, ++qq) { H[6 * rr + cc] = s[qq]; H[6 * cc + rr] = s[qq]; }
          for (int rr = 0; rr < 6; ++rr, ++qq) g[rr] = -s[qq]; }
        /* gauge: |t| is unobservable -> Marquardt damping plus a unit prior along t */
        for (int rr = 0; rr < 6; ++rr) H[6 * rr + rr] = H[6 * rr + rr] + lambda * H[6 * rr + rr];
        for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc) H[6 * (3 + rr) + 3 + cc] = H[6 * (3 + rr) + 3 + cc] + t[rr] * t[cc];
        double dl[6];
        if (chol6(H, g, dl)) { lambda = lambda * 10.0; if (lambda > 1e12) break; continue; }
        double Rn[9], tn[3];
        quat_right(R, dl, Rn);
        tn[0] = t[0] + dl[3]; tn[1] = t[1] + dl[4]; tn[2] = t[2] + dl[5];
        double nt = sqrt((tn[0] * tn[0] + tn[1] * tn[1]) + tn[2] * tn[2]);
        if (!(nt > 0.0)) { lambda = lambda * 10.0; if (lambda > 1e12) break; continue; }
        tn[0] = tn[0] / nt; tn[1] = tn[1] / nt; tn[2] = tn[2] / nt;
        double cn = emat_cost(x0, x1, idx, n, Rn, tn);
        double mx = 0.0;
        for (int k = 0; k < 6; ++k) { double v = dl[k] < 0.0 ? -dl[k] : dl[k]; if (v > mx) mx = v; }
        if (cn < cost) {
            double dec = cost - cn;
            memcpy(R, Rn, 72); memcpy(t, tn, 24);
            int done = (dec <= 1e-14 * cost);
            cost = cn;
            lambda = lambda * 0.1; if (lambda < 1e-12) lambda = 1e-12;
            if (done) break;
        } else { lambda = lambda * 10.0; if (lambda > 1e12) break; }
        if (mx < 1e-13) break;
    }
    free(w);
    return 0;
}

/* K-normalisation exactly as pose_solver.py:39-40: float32 keypoints minus / over K's entries -- numpy float32 arithmetic
 * for a float32 K, float64 arithmetic (keypoints widened) for the Map-free loader's float64 K */
void mfr_ref_normalize_points(const float *pts, int n, const void *K, int k_dtype, double *out)
{
    if (k_dtype == MFR_K_F32) {
        const float *k = (const float *)K;
        for (int i = 0; i < n; ++i) {
            float x = (pts[2 * i] - k[2]) / k[0], y = (pts[2 * i + 1] - k[5]) / k[4];
            out[2 * i] = (double)x; out[2 * i + 1] = (double)y;
        }
    } else {
        const double *k = (const double *)K;
        for (int i = 0; i < n; ++i) {
            out[2 * i] = ((double)pts[2 * i] - k[2]) / k[0];
            out[2 * i + 1] = ((double)pts[2 * i + 1] - k[5]) / k[4];
        }
    }
}

/* pose_solver.py:43: thr = PIX_THRESHOLD / np.mean([fx0, fy1, fy0, fx1]); the mean is taken in K's dtype */
double mfr_ref_emat_threshold(double pix_thr, const void *K0, const void *K1, int k_dtype)
{
    if (k_dtype == MFR_K_F32) {
        const float *a = (const float *)K0, *b = (const float *)K1;
        float m = (((a[0] + b[4]) + a[4]) + b[0]) / 4.0f;
        return pix_thr / (double)m;
    }
    const double *a = (const double *)K0, *b = (const double *)K1;
    double m = (((a[0] + b[4]) + a[4]) + b[0]) / 4.0;
    return pix_thr / m;
}

/* ------------------------------------------------------------------------------------------------------------------
 * MAGSAC++ quality + sigma-consensus++ local optimisation (score_method MFR_EMAT_SCORE_MAGSAC)
 * ------------------------------------------------------------------------------------------------------------------ */
#define MAGSAC_K 3.64            /* 0.99 quantile of the chi distribution with 4 degrees of freedom (the paper's k, OpenCV's sigma_quantile) */
#define MAGSAC_TILE 1024         /* loss sums: wave64 order inside a tile of 1024 points, tiles added in sequence */

/* Table of the normalised MAGSAC++ point loss and IRLS weight over u = r^2 / (k sigma_max)^2 in [0, 1], M intervals,
 * lut[2 j] = loss(u_j), lut[2 j + 1] = weight(u_j), u_j = j / M (linear interpolation in between; n = 4 degrees of freedom):
 *   x = r^2 / (2 sigma_max^2) = u k^2 / 2
 *   rho(x) / sigma_max^2 = 1/2 gamma(5/2, x) + x/2 (Gamma(3/2, x) - Gamma(3/2, k^2/2))           (paper eq. 6-8 up to a constant)
 *   loss(u)   = rho(x) / rho(k^2/2) - 1          in [-1, 0]: 0 for a point on the cut, -1 for a perfect fit
 *   weight(u) = (Gamma(3/2, x) - Gamma(3/2, k^2/2)) / (Gamma(3/2, 0) - Gamma(3/2, k^2/2))   = d rho / d r^2 up to a constant
 * closed forms: Gamma(1/2, x) = sqrt(pi) erfc(sqrt x), Gamma(a+1, x) = a Gamma(a, x) + x^a e^-x.
 * The table is the ONLY place libm (erfc, exp) enters; everything downstream is + - * / on its entries. */
void mfr_ref_magsac_lut(double *lut, int M)
{
    const double sqrt_pi = 1.7724538509055160273;
    const double xk = 0.5 * MAGSAC_K * MAGSAC_K;
    const double gk = 0.5 * sqrt_pi * erfc(sqrt(xk)) + sqrt(xk) * exp(-xk);                    /* Gamma(3/2, k^2/2) */
    const double norm = 0.75 * sqrt_pi - (1.5 * gk + xk * sqrt(xk) * exp(-xk));                /* gamma(5/2, k^2/2) */
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
}

typedef struct { const double *lut; int M; double cut, scale, thr2; } magsac_t;

static inline void magsac_lookup(const magsac_t *ms, double r2, double *loss, double *wgt)
{
    const double u = r2 * ms->scale;                 /* scale = M / cut, r2 < cut */
    int j = (int)u;
    if (j > ms->M - 1) j = ms->M - 1;
    const double f = u - (double)j;
    if (loss) *loss = ms->lut[2 * j] + f * (ms->lut[2 * j + 2] - ms->lut[2 * j]);
    if (wgt) *wgt = ms->lut[2 * j + 1] + f * (ms->lut[2 * j + 3] - ms->lut[2 * j + 1]);
}

static double butterfly64(double *a)
{
    for (int off = 32; off >= 1; off >>= 1) {
        double tmp[64];
        for (int l = 0; l < 64; ++l) tmp[l] = a[l] + a[l ^ off];
        memcpy(a, tmp, sizeof(tmp));
    }
    return a[0];
}

/* total loss (lower is better) and tentative inlier count (r^2 < thr^2) of one model */
static void magsac_score(const magsac_t *ms, const double *E, const double *x0, const double *x1, int n, double *loss, int *cnt)
{
    double L = 0.0; int c = 0;
    for (int base = 0; base < n; base += MAGSAC_TILE) {
        const int tn = (n - base < MAGSAC_TILE) ? n - base : MAGSAC_TILE;
        double acc[64];
        for (int l = 0; l < 64; ++l) acc[l] = 0.0;
        for (int i = 0; i < tn; ++i) {
            const int j = base + i;
            const double r2 = sampson2(E, x0[2 * j], x0[2 * j + 1], x1[2 * j], x1[2 * j + 1]);
            c += (r2 < ms->thr2);
            if (r2 < ms->cut) { double v; magsac_lookup(ms, r2, &v, NULL); acc[i & 63] = acc[i & 63] + v; }
        }
        L = L + butterfly64(acc);
    }
    *loss = L; *cnt = c;
}

/* two Newton-Schulz steps towards the orthogonal polar factor: R <- R (3 I - R^T R) / 2 (quadratic convergence; the inputs
 * are rotations up to ~1e-9, the outputs up to rounding) */
void mfr_ref_orthonormalize(double R[9])
{
    for (int it = 0; it < 2; ++it) {
        double S[9], Rn[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            const double g = (R[i] * R[j] + R[3 + i] * R[3 + j]) + R[6 + i] * R[6 + j];
            S[3 * i + j] = ((i == j) ? 3.0 : 0.0) - g;
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = 0.5 * ((R[3 * i] * S[j] + R[3 * i + 1] * S[3 + j]) + R[3 * i + 2] * S[6 + j]);
        memcpy(R, Rn, 72);
    }
}

#define MAGSAC_LO_START 100      /* USAC's max_iters_before_LO */
#define MAGSAC_LO_ITERS 20       /* re-weighting rounds per local optimisation */

/* sigma-consensus++: IRLS with the MAGSAC++ weights on (R, unit t); returns 0 and the optimised model / its score, or -1 */
static int magsac_lo(const magsac_t *ms, const double *x0, const double *x1, int n, const double *Ein,
                     double *Eout, double *loss_out, int *cnt_out)
{
    double R[9], Rb[9], t[3];
    if (mfr_ref_emat_decompose(Ein, R, Rb, t)) return -1;
    mfr_ref_orthonormalize(R);
    double E[9]; skew_mul(t, R, E);
    double loss; int cnt;
    magsac_score(ms, E, x0, x1, n, &loss, &cnt);
    if (!(loss == loss)) return -1;
    double lambda = 1e-3;
    wacc_t *w = (wacc_t *)malloc(sizeof(wacc_t));
    for (int it = 0; it < MAGSAC_LO_ITERS; ++it) {
        memset(w, 0, sizeof(*w));
        for (int i = 0; i < n; ++i) {
            double a = x0[2 * i], b = x0[2 * i + 1], c = x1[2 * i], d = x1[2 * i + 1];
            double Ex0 = (E[0] * a + E[1] * b) + E[2], Ex1 = (E[3] * a + E[4] * b) + E[5], Ex2 = (E[6] * a + E[7] * b) + E[8];
            double Et0 = (E[0] * c + E[3] * d) + E[6], Et1 = (E[1] * c + E[4] * d) + E[7];
            double num = (c * Ex0 + d * Ex1) + Ex2;
            double den = ((Ex0 * Ex0 + Ex1 * Ex1) + Et0 * Et0) + Et1 * Et1;
            double r2 = (num * num) / den;
            if (!(r2 < ms->cut)) continue;
            double pw; magsac_lookup(ms, r2, NULL, &pw);
            double wgt = 1.0 / sqrt(den);
            double q[3] = { c, d, 1.0 }, p[3] = { a, b, 1.0 };
            double txq[3] = { t[1] * q[2] - t[2] * q[1], t[2] * q[0] - t[0] * q[2], t[0] * q[1] - t[1] * q[0] };
            double u[3] = { -((R[0] * txq[0] + R[3] * txq[1]) + R[6] * txq[2]),
                            -((R[1] * txq[0] + R[4] * txq[1]) + R[7] * txq[2]),
                            -((R[2] * txq[0] + R[5] * txq[1]) + R[8] * txq[2]) };
            double Rp[3] = { (R[0] * p[0] + R[1] * p[1]) + R[2] * p[2], (R[3] * p[0] + R[4] * p[1]) + R[5] * p[2],
                             (R[6] * p[0] + R[7] * p[1]) + R[8] * p[2] };
            double J[6];
            J[0] = (p[1] * u[2] - p[2] * u[1]) * wgt; J[1] = (p[2] * u[0] - p[0] * u[2]) * wgt; J[2] = (p[0] * u[1] - p[1] * u[0]) * wgt;
            J[3] = (Rp[1] * q[2] - Rp[2] * q[1]) * wgt; J[4] = (Rp[2] * q[0] - Rp[0] * q[2]) * wgt; J[5] = (Rp[0] * q[1] - Rp[1] * q[0]) * wgt;
            double r = num * wgt;
            double *acc = w->a[i % EM_SEL_LANES];
            int qq = 0;
            for (int rr = 0; rr < 6; ++rr) { double wj = pw * J[rr]; for (int cc = rr; cc < 6; ++cc, ++qq) acc[qq] = acc[qq] + wj * J[cc]; }
            for (int rr = 0; rr < 6; ++rr, ++qq) acc[qq] = acc[qq] + (pw * J[rr]) * r;
        }
        double s[27]; wacc_finish(w, 27, s);
        double H[36], g[6];
        { int qq = 0;
          for (int rr = 0; rr < 6; ++rr) for (int cc = rr; cc < 6; ++cc, ++qq) { H[6 * rr + cc] = s[qq]; H[6 * cc + rr] = s[qq]; }
          for (int rr = 0; rr < 6; ++rr, ++qq) g[rr] = -s[qq]; }
        for (int rr = 0; rr < 6; ++rr) H[6 * rr + rr] = H[6 * rr + rr] + lambda * H[6 * rr + rr];
        for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc) H[6 * (3 + rr) + 3 + cc] = H[6 * (3 + rr) + 3 + cc] + t[rr] * t[cc];
        double dl[6];
        if (chol6(H, g, dl)) { lambda = lambda * 10.0; if (lambda > 1e12) break; continue; }
        double Rn[9], tn[3], En[9];
        quat_right(R, dl, Rn);
        tn[0] = t[0] + dl[3]; tn[1] = t[1] + dl[4]; tn[2] = t[2] + dl[5];
        double nt = sqrt((tn[0] * tn[0] + tn[1] * tn[1]) + tn[2] * tn[2]);
        if (!(nt > 0.0)) { lambda = lambda * 10.0; if (lambda > 1e12) break; continue; }
        tn[0] = tn[0] / nt; tn[1] = tn[1] / nt; tn[2] = tn[2] / nt;
        skew_mul(tn, Rn, En);
        double ln; int cn;
        magsac_score(ms, En, x0, x1, n, &ln, &cn);
        double mx = 0.0;
        for (int k = 0; k < 6; ++k) { double v = dl[k] < 0.0 ? -dl[k] : dl[k]; if (v > mx) mx = v; }
        if (ln < loss) {
            double dec = loss - ln, mag = loss < 0.0 ? -loss : loss;
            memcpy(R, Rn, 72); memcpy(t, tn, 24); memcpy(E, En, 72);
            int done = (dec <= 1e-12 * mag);
            loss = ln; cnt = cn;
            lambda = lambda * 0.1; if (lambda < 1e-12) lambda = 1e-12;
            if (done) break;
        } else { lambda = lambda * 10.0; if (lambda > 1e12) break; }
        if (mx < 1e-13) break;
    }
    free(w);
    memcpy(Eout, E, 72); *loss_out = loss; *cnt_out = cnt;
    return 0;
}

/* one hypothesis: best of the <= 10 five-point models -- most inliers (COUNT) or smallest MAGSAC++ loss (MAGSAC), first on ties.
 * Returns the tentative inlier count of that model; *loss_best its loss (MAGSAC only; 0 = no model). */
static int emat_hypothesis(const double *x0, const double *x1, int n, const int *s, double thr2, const magsac_t *ms,
                           double *Ebest, double *loss_best)
{
    double a[10], b[10], Es[90];
    for (int k = 0; k < 5; ++k) { a[2 * k] = x0[2 * s[k]]; a[2 * k + 1] = x0[2 * s[k] + 1]; b[2 * k] = x1[2 * s[k]]; b[2 * k + 1] = x1[2 * s[k] + 1]; }
    int ns = mfr_ref_fivept(a, b, Es);
    int best = 0;
    if (ms) {
        double bl = 0.0;
        for (int m = 0; m < ns; ++m) {
            double l; int c;
            magsac_score(ms, Es + 9 * m, x0, x1, n, &l, &c);
            if (l < bl) { bl = l; best = c; memcpy(Ebest, Es + 9 * m, 72); }
        }
        *loss_best = bl;
        return best;
    }
    for (int m = 0; m < ns; ++m) {
        int cnt = 0;
        for (int i = 0; i < n; ++i) cnt += (sampson2(Es + 9 * m, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]) <= thr2);
        if (cnt > best) { best = cnt; memcpy(Ebest, Es + 9 * m, 72); }
    }
    return best;
}

/* cv.recoverPose (pose_solver.py:56-60): 4 decompositions, keep the one with most masked points in front of both cameras */
static int recover_pose(const double *E, const double *x0, const double *x1, const int32_t *idx, int m, double *Rb, double *tb)
{
    double Ra[9], Rc[9], tu[3];
    if (mfr_ref_emat_decompose(E, Ra, Rc, tu)) return -1;
    int bestc = -1;
    for (int c = 0; c < 4; ++c) {
        const double *Rk = (c < 2) ? Ra : Rc;
        double tk[3] = { (c & 1) ? -tu[0] : tu[0], (c & 1) ? -tu[1] : tu[1], (c & 1) ? -tu[2] : tu[2] };
        int cnt = 0;
        for (int q = 0; q < m; ++q) { int i = idx[q]; cnt += cheirality(Rk, tk, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]); }
        if (cnt > bestc) { bestc = cnt; memcpy(Rb, Rk, 72); memcpy(tb, tk, 24); }
    }
    return bestc;
}

/* EssentialMatrixSolver.estimate_pose (pose_solver.py:29-61).  mask_out = cheirality-filtered
 * inliers (what self.mask holds after the recoverPose loop, quirk Q7), n_inl = their count.
 * score_method MFR_EMAT_SCORE_MAGSAC needs lut = mfr_ref_magsac_lut(.., lut_m); losses [max_iters] (may be NULL) receives the
 * per-hypothesis loss of the iterations that ran (0 beyond). */
int mfr_ref_emat_solve(const float *pts0, const float *pts1, int n, const void *K0, const void *K1, int k_dtype,
                       double pix_thr, double conf, int max_iters, uint64_t seed, uint64_t pair_id,
                       int score_method, const double *lut, int lut_m, double max_thr_ratio,
                       double R[9], double t[3], uint8_t *mask_out, int *n_inl,
                       int *best_iter, int *iters_run, int32_t *counts, double *losses, uint8_t *ransac_mask, int *lo_runs)
{
    for (int i = 0; i < 9; ++i) R[i] = NAN;
    for (int i = 0; i < 3; ++i) t[i] = NAN;
    *n_inl = 0; if (best_iter) *best_iter = -1; if (iters_run) *iters_run = 0; if (lo_runs) *lo_runs = 0;
    if (mask_out) memset(mask_out, 0, (size_t)(n > 0 ? n : 0));
    if (ransac_mask) memset(ransac_mask, 0, (size_t)(n > 0 ? n : 0));
    if (n < 5) return MFR_ST_TOO_FEW;                                        /* :32-33 */
    if (max_iters < 1) max_iters = 1;
    const int magsac = (score_method == MFR_EMAT_SCORE_MAGSAC);
    if (magsac && (!lut || lut_m < 2 || !(max_thr_ratio >= 1.0))) return -1;
    double *x0 = (double *)malloc(sizeof(double) * 2 * (size_t)n), *x1 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    mfr_ref_normalize_points(pts0, n, K0, k_dtype, x0);                      /* :39 */
    mfr_ref_normalize_points(pts1, n, K1, k_dtype, x1);                      /* :40 */
    double thr = mfr_ref_emat_threshold(pix_thr, K0, K1, k_dtype), thr2 = thr * thr;  /* :43 */
    magsac_t msv, *ms = NULL;
    if (magsac) {
        msv.lut = lut; msv.M = lut_m; msv.thr2 = thr2;
        msv.cut = (max_thr_ratio * max_thr_ratio) * thr2;
        msv.scale = (double)lut_m / msv.cut;
        ms = &msv;
    }
    double Eb[9], Eh[9];
    int best = magsac ? 0 : 4, bit = -1, niters = max_iters, it = 0, nlo = 0, best_is_lo = 0;
    double best_loss = 0.0;
    if (n == 5) {
        int s[5] = { 0, 1, 2, 3, 4 };
        double l = 0.0;
        int c = emat_hypothesis(x0, x1, n, s, thr2, ms, Eb, &l);
        it = 1;
        if (counts) counts[0] = c;
        if (losses) losses[0] = l;
        if (magsac ? (l < 0.0) : (c > 0)) { best = c; bit = 0; best_loss = l; }
        if (counts) for (int k = 1; k < max_iters; ++k) counts[k] = -1;
        if (losses) for (int k = 1; k < max_iters; ++k) losses[k] = 0.0;
    } else {
        for (it = 0; it < niters; ++it) {
            int s[5];
            double l = 0.0;
            mfr_ref_sample_distinct(seed, pair_id, (uint32_t)it, n, 5, s);
            int cnt = emat_hypothesis(x0, x1, n, s, thr2, ms, Eh, &l);
            if (counts) counts[it] = cnt;
            if (losses) losses[it] = l;
            if (magsac ? (l < best_loss) : (cnt > best)) {
                best = cnt; bit = it; memcpy(Eb, Eh, 72); best_loss = l; best_is_lo = 0;
                if (magsac && it >= MAGSAC_LO_START) {
                    double El[9], ll; int cl;
                    ++nlo;
                    if (magsac_lo(ms, x0, x1, n, Eb, El, &ll, &cl) == 0 && ll < best_loss) {
                        memcpy(Eb, El, 72); best_loss = ll; best = cl; best_is_lo = 1;
                    }
                }
                niters = mfr_ref_update_num_iters(conf, (double)(n - best) / (double)n, 5, niters);
            }
        }
        if (counts) for (int k = it; k < max_iters; ++k) counts[k] = -1;
        if (losses) for (int k = it; k < max_iters; ++k) losses[k] = 0.0;
    }
    if (iters_run) *iters_run = it;
    if (best_iter) *best_iter = bit;
    int st = MFR_ST_OK;
    if (bit < 0) st = MFR_ST_NO_MODEL;                                        /* E is None -> :50-51 */
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    uint8_t *rm = (uint8_t *)calloc((size_t)n, 1);
    double Rb[9], tb[3];
    if (st == MFR_ST_OK && magsac) {
        if (!best_is_lo && n > 5) {                                           /* the winner never went through the local optimisation */
            double El[9], ll; int cl;
            ++nlo;
            if (magsac_lo(ms, x0, x1, n, Eb, El, &ll, &cl) == 0 && ll < best_loss) { memcpy(Eb, El, 72); best_loss = ll; best = cl; }
        }
        int m = 0;
        for (int i = 0; i < n; ++i) {
            rm[i] = (uint8_t)(sampson2(Eb, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]) < thr2);
            if (rm[i]) idx[m++] = i;
        }
        if (recover_pose(Eb, x0, x1, idx, m, Rb, tb) <= 0) st = MFR_ST_NO_MODEL;   /* n == 0 -> ret stays NaN (:54-60) */
    } else if (st == MFR_ST_OK) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            rm[i] = (uint8_t)(sampson2(Eb, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]) <= thr2);
            if (rm[i]) idx[m++] = i;
        }
        /* recoverPose: 4 candidates, keep the one with most points in front of both cameras (:56-60) */
        if (recover_pose(Eb, x0, x1, idx, m, Rb, tb) <= 0) st = MFR_ST_NO_MODEL;
        if (st == MFR_ST_OK && n > 5) {
            /* polish on the RANSAC inliers (stand-in for USAC LO + final polisher) */
            double Rr[9], tr[3];
            memcpy(Rr, Rb, 72); memcpy(tr, tb, 24);
            if (mfr_ref_emat_refine(x0, x1, idx, m, 20, Rr, tr) == 0) {
                double Er[9]; skew_mul(tr, Rr, Er);
                int m2 = 0;
                for (int i = 0; i < n; ++i) m2 += (sampson2(Er, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]) <= thr2);
                if (m2 >= m) {                                               /* keep the polish only if it is no worse */
                    memcpy(Rb, Rr, 72); memcpy(tb, tr, 24);
                    for (int i = 0; i < n; ++i) rm[i] = (uint8_t)(sampson2(Er, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]) <= thr2);
                }
            }
        }
    }
    if (st == MFR_ST_OK) {
        mfr_ref_orthonormalize(Rb);                                           /* Horn's R inherits E's distance from the essential manifold */
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            int in = rm[i] && cheirality(Rb, tb, x0[2 * i], x0[2 * i + 1], x1[2 * i], x1[2 * i + 1]);
            if (mask_out) mask_out[i] = (uint8_t)in;
            cnt += in;
        }
        if (cnt <= 0) st = MFR_ST_NO_MODEL;
        else { *n_inl = cnt; memcpy(R, Rb, 72); memcpy(t, tb, 24); }
        if (ransac_mask) memcpy(ransac_mask, rm, (size_t)n);
    }
    if (st != MFR_ST_OK && mask_out) memset(mask_out, 0, (size_t)n);
    if (lo_runs) *lo_runs = nlo;
    free(x0); free(x1); free(idx); free(rm);
    return st;
}
