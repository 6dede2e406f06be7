// emat_dev.h -- per-lane essential-matrix geometry for the E-mat RANSAC kernels (gfx950).
// Device statement of EssentialMatrixSolver's arithmetic (lib/models/matching/pose_solver.py:29-61;
// cv.findEssentialMat / cv.recoverPose restated from the published algorithms, see emat.hip).
// Same FP contract as geom_dev.h (binary64, + - * / sqrt only, no FMA contraction, fixed order).
#pragma once
#include "geom_dev.h"

namespace mfr {

__device__ static const int IDX11[4][4] = { {0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9} };
__device__ static const int IDX21[10][4] = { {0, 1, 2, 10}, {1, 3, 4, 11}, {2, 4, 5, 12}, {3, 6, 7, 13}, {4, 7, 8, 14},
                                             {5, 8, 9, 15}, {10, 11, 12, 16}, {11, 13, 14, 17}, {12, 14, 15, 18},
                                             {16, 17, 18, 19} };
__device__ static const int NPERM[20] = { 0, 6, 1, 3, 2, 10, 7, 13, 4, 11, 5, 12, 16, 8, 14, 17, 9, 15, 18, 19 };

MFR_DEV void p_mul11(const double *a, const double *b, double *o)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) o[IDX11[i][j]] = o[IDX11[i][j]] + a[i] * b[j];
}
MFR_DEV void p_mul21(const double *a, const double *b, double *o)
{
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) o[IDX21[i][j]] = o[IDX21[i][j]] + a[i] * b[j];
}

// Nister's 5-point solver.  x0, x1: 5 normalised points each; up to 10 E (row-major, unit Frobenius
// norm), x1^T E x0 = 0.
MFR_DEV_NOINLINE int fivept(const double *x0, const double *x1, double *Es)
{
    double A[5][9];
    for (int i = 0; i < 5; ++i) {
        const double a = x0[2 * i], b = x0[2 * i + 1], c = x1[2 * i], d = x1[2 * i + 1];
        A[i][0] = c * a; A[i][1] = c * b; A[i][2] = c; A[i][3] = d * a; A[i][4] = d * b; A[i][5] = d;
        A[i][6] = a; A[i][7] = b; A[i][8] = 1.0;
    }
    int colp[9];
    for (int j = 0; j < 9; ++j) colp[j] = j;
    for (int r = 0; r < 5; ++r) {
        int pr = r, pc = r;
        double best = -1.0;
        for (int i = r; i < 5; ++i)
            for (int j = r; j < 9; ++j) {
                const double v = A[i][j] < 0.0 ? -A[i][j] : A[i][j];
                if (v > best) { best = v; pr = i; pc = j; }
            }
        if (!(best > 1e-300)) return 0;
        if (pr != r)
            for (int j = 0; j < 9; ++j) { const double tmp = A[r][j]; A[r][j] = A[pr][j]; A[pr][j] = tmp; }
        if (pc != r) {
            for (int i = 0; i < 5; ++i) { const double tmp = A[i][r]; A[i][r] = A[i][pc]; A[i][pc] = tmp; }
            const int ti = colp[r]; colp[r] = colp[pc]; colp[pc] = ti;
        }
        const double inv = 1.0 / A[r][r];
        for (int j = 0; j < 9; ++j) A[r][j] = A[r][j] * inv;
        for (int i = 0; i < 5; ++i)
            if (i != r) {
                const double f = A[i][r];
                for (int j = 0; j < 9; ++j) A[i][j] = A[i][j] - f * A[r][j];
            }
    }
    double Ep[9][4];
    for (int k = 0; k < 4; ++k) {
        double v[9];
        for (int j = 0; j < 9; ++j) v[j] = 0.0;
        v[5 + k] = 1.0;
        for (int r = 0; r < 5; ++r) v[r] = -A[r][5 + k];
        for (int j = 0; j < 9; ++j) Ep[colp[j]][k] = v[j];
    }
    double M[10][20];
    {
        double C[10][20];
        for (int r = 0; r < 10; ++r)
            for (int c = 0; c < 20; ++c) C[r][c] = 0.0;
        {
            double m[10], neg[4];
#define MFR_MINOR(a, b, c, d)                                  \
    do {                                                       \
        for (int q = 0; q < 10; ++q) m[q] = 0.0;               \
        p_mul11(Ep[a], Ep[b], m);                              \
        for (int q = 0; q < 4; ++q) neg[q] = -Ep[c][q];        \
        p_mul11(neg, Ep[d], m);                                \
    } while (0)
            MFR_MINOR(4, 8, 5, 7); p_mul21(m, Ep[0], C[0]);
            MFR_MINOR(5, 6, 3, 8); p_mul21(m, Ep[1], C[0]);
            MFR_MINOR(3, 7, 4, 6); p_mul21(m, Ep[2], C[0]);
#undef MFR_MINOR
        }
        {
            double EEt[3][3][10];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    for (int q = 0; q < 10; ++q) EEt[i][j][q] = 0.0;
                    for (int k = 0; k < 3; ++k) p_mul11(Ep[3 * i + k], Ep[3 * j + k], EEt[i][j]);
                }
            double tr[10];
            for (int q = 0; q < 10; ++q) tr[q] = (EEt[0][0][q] + EEt[1][1][q]) + EEt[2][2][q];
            for (int i = 0; i < 3; ++i)
                for (int q = 0; q < 10; ++q) EEt[i][i][q] = EEt[i][i][q] - 0.5 * tr[q];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    for (int k = 0; k < 3; ++k) p_mul21(EEt[i][k], Ep[3 * k + j], C[1 + 3 * i + j]);
        }
        for (int r = 0; r < 10; ++r)
            for (int c = 0; c < 20; ++c) M[r][c] = C[r][NPERM[c]];
    }
    for (int c = 0; c < 10; ++c) {
        int pr = c;
        double best = -1.0;
        for (int i = c; i < 10; ++i) {
            const double v = M[i][c] < 0.0 ? -M[i][c] : M[i][c];
            if (v > best) { best = v; pr = i; }
        }
        if (!(best > 1e-300)) return 0;
        if (pr != c)
            for (int j = 0; j < 20; ++j) { const double tmp = M[c][j]; M[c][j] = M[pr][j]; M[pr][j] = tmp; }
        const double inv = 1.0 / M[c][c];
        for (int j = 0; j < 20; ++j) M[c][j] = M[c][j] * inv;
        for (int i = 0; i < 10; ++i)
            if (i != c) {
                const double f = M[i][c];
                for (int j = 0; j < 20; ++j) M[i][j] = M[i][j] - f * M[c][j];
            }
    }
    double Bx[3][4], By[3][4], B1[3][5];
    for (int r = 0; r < 3; ++r) {
        const double *e = M[4 + 2 * r], *f = M[5 + 2 * r];
        Bx[r][0] = e[12]; Bx[r][1] = e[11] - f[12]; Bx[r][2] = e[10] - f[11]; Bx[r][3] = -f[10];
        By[r][0] = e[15]; By[r][1] = e[14] - f[15]; By[r][2] = e[13] - f[14]; By[r][3] = -f[13];
        B1[r][0] = e[19]; B1[r][1] = e[18] - f[19]; B1[r][2] = e[17] - f[18]; B1[r][3] = e[16] - f[17]; B1[r][4] = -f[16];
    }
    double P[11];
    for (int q = 0; q < 11; ++q) P[q] = 0.0;
    {
        double c0[8], c1[8], c2[7];
        for (int q = 0; q < 8; ++q) { c0[q] = 0.0; c1[q] = 0.0; }
        for (int q = 0; q < 7; ++q) c2[q] = 0.0;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 5; ++b) {
                c0[a + b] = c0[a + b] + (By[1][a] * B1[2][b] - B1[1][b] * By[2][a]);
                c1[a + b] = c1[a + b] + (Bx[1][a] * B1[2][b] - B1[1][b] * Bx[2][a]);
            }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) c2[a + b] = c2[a + b] + (Bx[1][a] * By[2][b] - By[1][a] * Bx[2][b]);
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 8; ++b) P[a + b] = P[a + b] + (Bx[0][a] * c0[b] - By[0][a] * c1[b]);
        for (int a = 0; a < 5; ++a)
            for (int b = 0; b < 7; ++b) P[a + b] = P[a + b] + B1[0][a] * c2[b];
    }
    double roots[10];
    const int nr = poly_real_roots<10>(P, 10, roots);
    int ns = 0;
    for (int r = 0; r < nr; ++r) {
        const double z = roots[r];
        double bx[3], by[3], b1[3];
        for (int k = 0; k < 3; ++k) {
            bx[k] = ((Bx[k][3] * z + Bx[k][2]) * z + Bx[k][1]) * z + Bx[k][0];
            by[k] = ((By[k][3] * z + By[k][2]) * z + By[k][1]) * z + By[k][0];
            b1[k] = (((B1[k][4] * z + B1[k][3]) * z + B1[k][2]) * z + B1[k][1]) * z + B1[k][0];
        }
        double v[3] = { 0.0, 0.0, 0.0 }, bestw = -1.0;
        for (int a = 0; a < 3; ++a) {
            const int p = a, q = (a + 1) % 3;
            const double w0 = by[p] * b1[q] - b1[p] * by[q];
            const double w1 = b1[p] * bx[q] - bx[p] * b1[q];
            const double w2 = bx[p] * by[q] - by[p] * bx[q];
            const double aw = w2 < 0.0 ? -w2 : w2;
            if (aw > bestw) { bestw = aw; v[0] = w0; v[1] = w1; v[2] = w2; }
        }
        if (!(bestw > 0.0)) continue;
        const double x = v[0] / v[2], y = v[1] / v[2];
        double *E = Es + 9 * ns, nn = 0.0;
        for (int e = 0; e < 9; ++e) {
            E[e] = ((x * Ep[e][0] + y * Ep[e][1]) + z * Ep[e][2]) + Ep[e][3];
            nn = nn + E[e] * E[e];
        }
        if (!(nn > 0.0) || !(nn < 1e300)) continue;
        const double s = 1.0 / sqrt(nn);
        for (int e = 0; e < 9; ++e) E[e] = E[e] * s;
        ++ns;
    }
    return ns;
}

MFR_DEV double sampson2(const double *E, double a, double b, double c, double d)
{
    const double Ex0 = (E[0] * a + E[1] * b) + E[2], Ex1 = (E[3] * a + E[4] * b) + E[5], Ex2 = (E[6] * a + E[7] * b) + E[8];
    const double Et0 = (E[0] * c + E[3] * d) + E[6], Et1 = (E[1] * c + E[4] * d) + E[7];
    const double num = (c * Ex0 + d * Ex1) + Ex2;
    const double den = ((Ex0 * Ex0 + Ex1 * Ex1) + Et0 * Et0) + Et1 * Et1;
    return (num * num) / den;
}

MFR_DEV void skew_mul(const double *t, const double *R, double *E)
{
    for (int j = 0; j < 3; ++j) {
        E[j]     = t[1] * R[6 + j] - t[2] * R[3 + j];
        E[3 + j] = t[2] * R[j]     - t[0] * R[6 + j];
        E[6 + j] = t[0] * R[3 + j] - t[1] * R[j];
    }
}

// Horn 1990 closed-form decomposition (stands in for the SVD inside cv::recoverPose)
MFR_DEV int emat_decompose(const double *E, double *Ra, double *Rb, double *tu)
{
    double EEt[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            EEt[3 * i + j] = (E[3 * i] * E[3 * j] + E[3 * i + 1] * E[3 * j + 1]) + E[3 * i + 2] * E[3 * j + 2];
    const double htr = 0.5 * ((EEt[0] + EEt[4]) + EEt[8]);
    double bb[9];
    for (int i = 0; i < 9; ++i) bb[i] = -EEt[i];
    bb[0] = bb[0] + htr; bb[4] = bb[4] + htr; bb[8] = bb[8] + htr;
    // column k of bb with the largest diagonal entry -- by selects, not by an index into the local array (a run-time index puts bb in scratch)
    int k = 0;
    double dk = bb[0];
    if (bb[4] > dk) { k = 1; dk = bb[4]; }
    if (bb[8] > dk) { k = 2; dk = bb[8]; }
    if (!(dk > 0.0)) return -1;
    const double s = sqrt(dk);
    const double c0 = k == 0 ? bb[0] : k == 1 ? bb[1] : bb[2], c1 = k == 0 ? bb[3] : k == 1 ? bb[4] : bb[5], c2 = k == 0 ? bb[6] : k == 1 ? bb[7] : bb[8];
    const double b[3] = { c0 / s, c1 / s, c2 / s };
    double C[9];
    C[0] = E[4] * E[8] - E[5] * E[7]; C[1] = -(E[3] * E[8] - E[5] * E[6]); C[2] = E[3] * E[7] - E[4] * E[6];
    C[3] = -(E[1] * E[8] - E[2] * E[7]); C[4] = E[0] * E[8] - E[2] * E[6]; C[5] = -(E[0] * E[7] - E[1] * E[6]);
    C[6] = E[1] * E[5] - E[2] * E[4]; C[7] = -(E[0] * E[5] - E[2] * E[3]); C[8] = E[0] * E[4] - E[1] * E[3];
    double bE[9];
    skew_mul(b, E, bE);
    const double b2 = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    for (int i = 0; i < 9; ++i) { Ra[i] = (C[i] - bE[i]) / b2; Rb[i] = (C[i] + bE[i]) / b2; }
    const double nb = sqrt(b2);
    tu[0] = b[0] / nb; tu[1] = b[1] / nb; tu[2] = b[2] / nb;
    return 0;
}

// closest-point depths along both rays > 0 (cheirality vote of cv::recoverPose)
MFR_DEV bool cheirality(const double *R, const double *t, double a, double b, double c, double d)
{
    const double p[3] = { (R[0] * a + R[1] * b) + R[2], (R[3] * a + R[4] * b) + R[5], (R[6] * a + R[7] * b) + R[8] };
    const double q[3] = { c, d, 1.0 };
    const double pp = (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2], qq = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    const double pq = (p[0] * q[0] + p[1] * q[1]) + p[2] * q[2];
    const double pt = (p[0] * t[0] + p[1] * t[1]) + p[2] * t[2], qt = (q[0] * t[0] + q[1] * t[1]) + q[2] * t[2];
    const double det = pp * qq - pq * pq;
    if (!(det > 1e-18 * pp * qq)) return false;
    const double l0 = (pq * qt - qq * pt) / det;
    const double l1 = (pp * qt - pq * pt) / det;
    return (l0 > 0.0) && (l1 > 0.0);
}

}  // namespace mfr
