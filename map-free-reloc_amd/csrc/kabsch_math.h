/* kabsch_math.h -- the arithmetic of kabsch.hip, written so that the same lines compile as device code (hipcc) and as plain C99
 * (gcc: tests/test_kabsch_host.py builds this header on the host and checks it against torch's SVD-based Kabsch and its
 * autograd gradient).  Binary64 throughout; matrices are row-major flat arrays. */
#ifndef MFR_KABSCH_MATH_H
#define MFR_KABSCH_MATH_H
#include <math.h>

#ifndef KB_FN
#define KB_FN static
#endif

/* cyclic Jacobi on a symmetric n x n matrix (n <= 4): A -> diagonal (eigenvalues), V = eigenvectors (columns) */
KB_FN void kb_jacobi_sym(double *A, double *V, int n)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int p = 0; p < n - 1; ++p)
            for (int r = p + 1; r < n; ++r) {
                const double apq = A[p * n + r];
                if (apq == 0.0) continue;
                const double theta = (A[r * n + r] - A[p * n + p]) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { const double a = A[k * n + p], b = A[k * n + r]; A[k * n + p] = c * a - s * b; A[k * n + r] = s * a + c * b; }
                for (int k = 0; k < n; ++k) { const double a = A[p * n + k], b = A[r * n + k]; A[p * n + k] = c * a - s * b; A[r * n + k] = s * a + c * b; }
                for (int k = 0; k < n; ++k) { const double a = V[k * n + p], b = V[k * n + r]; V[k * n + p] = c * a - s * b; V[k * n + r] = s * a + c * b; }
            }
}

/* Horn: the proper rotation R maximising tr(R H) for H = sum_k a_k b_k^T (rows a = source, b = target): b ~ R a */
KB_FN void kb_horn_rotation(const double *S, double *R)
{
    double N[16], V[16];
    N[0] = S[0] + S[4] + S[8];
    N[1] = S[5] - S[7]; N[2] = S[6] - S[2]; N[3] = S[1] - S[3];
    N[5] = S[0] - S[4] - S[8]; N[6] = S[1] + S[3]; N[7] = S[6] + S[2];
    N[10] = -S[0] + S[4] - S[8]; N[11] = S[5] + S[7];
    N[15] = -S[0] - S[4] + S[8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < i; ++j) N[i * 4 + j] = N[j * 4 + i];
    kb_jacobi_sym(N, V, 4);
    int b = 0;
    for (int i = 1; i < 4; ++i) if (N[i * 5] > N[b * 5]) b = i;
    double nn = sqrt(V[b] * V[b] + V[4 + b] * V[4 + b] + V[8 + b] * V[8 + b] + V[12 + b] * V[12 + b]);
    if (!(nn > 0.0)) nn = 1.0;
    const double w = V[b] / nn, x = V[4 + b] / nn, y = V[8 + b] / nn, z = V[12 + b] / nn;
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
}

/* C = op(A) op(B) for 3x3 row-major; ta / tb: use the transpose */
KB_FN void kb_mm3(const double *A, int ta, const double *B, int tb, double *C)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double a = 0.0;
        for (int k = 0; k < 3; ++k) a += (ta ? A[k * 3 + i] : A[i * 3 + k]) * (tb ? B[j * 3 + k] : B[k * 3 + j]);
        C[i * 3 + j] = a;
    }
}

/* G = dL/dR -> gH = dL/dH, for R = kb_horn_rotation(H) */
KB_FN void kb_rotation_backward(const double *H, const double *G, double *gH)
{
    double R[9], S[9], U[9], P[9], A[9], T[9], Ap[9], W[9], RW[9];
    kb_horn_rotation(H, R);
    kb_mm3(R, 1, H, 1, S);                                    /* S = R^T H^T, symmetric up to round-off */
    for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) { const double m = 0.5 * (S[i * 3 + j] + S[j * 3 + i]); S[i * 3 + j] = m; S[j * 3 + i] = m; }
    kb_jacobi_sym(S, U, 3);
    const double s[3] = { S[0], S[4], S[8] };
    kb_mm3(R, 1, G, 0, P);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i * 3 + j] = 0.5 * (P[i * 3 + j] - P[j * 3 + i]);
    kb_mm3(A, 0, U, 0, T);
    kb_mm3(U, 1, T, 0, Ap);                                   /* A' = U^T A U */
    const double scale = fabs(s[0]) + fabs(s[1]) + fabs(s[2]);
    const double fl = 1e-12 * (scale > 0.0 ? scale : 1.0);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double d = s[i] + s[j];
        if (fabs(d) < fl) d = d < 0.0 ? -fl : fl;             /* degenerate configurations: bounded, like a clamped SVD gradient */
        Ap[i * 3 + j] = (i == j) ? 0.0 : Ap[i * 3 + j] / d;
    }
    kb_mm3(Ap, 0, U, 1, T);
    kb_mm3(U, 0, T, 0, W);                                    /* W = U W' U^T */
    kb_mm3(R, 0, W, 0, RW);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gH[j * 3 + i] = 2.0 * RW[i * 3 + j];     /* (2 R W)^T */
}
#endif
