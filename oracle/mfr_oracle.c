/*
 * oracle/mfr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mfr_oracle.h).
 *
 * Scalar, straight-line C restatement of the pose-solver leg of the reference
 * (lib/models/matching/pose_solver.py) plus the OpenCV routines it calls
 * (restated from the published algorithms; OpenCV is not available offline).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include "mfr_oracle.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) -- counter-based RNG shared with the */
/* device kernels.  The reference has no seed knob: OpenCV's    */
/* fixed RNG((uint64)-1) is replaced by (seed, pair_id, iteration) counters. */
/* ------------------------------------------------------------------------ */
void mfr_ref_philox4x32_10(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* k distinct indices in [0,n), draw order preserved (stands in for OpenCV's
 * RANSACPointSetRegistrator::getSubset, ptsetreg.cpp).  Word j of the stream
 * for (seed,pair,iter) is philox(ctr={iter, j/4, pair_lo, pair_hi}, key=seed)[j%4];
 * v = mulhi32(word, n-j) is mapped to the v-th not-yet-chosen index. */
void mfr_ref_sample_distinct(uint64_t seed, uint64_t pair_id, uint32_t iter, int n, int k, int *out)
{
    int sorted[8];
    uint32_t w[4];
    for (int j = 0; j < k; ++j) {
        if ((j & 3) == 0) {
            uint32_t ctr[4] = { iter, (uint32_t)(j >> 2), (uint32_t)pair_id, (uint32_t)(pair_id >> 32) };
            mfr_ref_philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        }
        int v = (int)(((uint64_t)w[j & 3] * (uint64_t)(uint32_t)(n - j)) >> 32);
        int pos = 0;
        while (pos < j && v >= sorted[pos]) { ++v; ++pos; }
        for (int q = j; q > pos; --q) sorted[q] = sorted[q - 1];
        sorted[pos] = v;
        out[j] = v;
    }
}

/* ------------------------------------------------------------------------ */
/* log() from + - * / only, so host and device agree bit-for-bit.            */
/* ------------------------------------------------------------------------ */
double mfr_ref_det_log(double x)
{
    uint64_t b; memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) { x = x * 18014398509481984.0; memcpy(&b, &x, 8); e = (int)((b >> 52) & 0x7ff) - 54; }
    e -= 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m; memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double acc = 0.0;
    for (int k = 17; k >= 0; --k) acc = acc * s2 + 1.0 / (double)(2 * k + 1);
    return 2.0 * s * acc + (double)e * 0.6931471805599453;
}

/* OpenCV RANSACUpdateNumIters (calib3d/src/ptsetreg.cpp) */
int mfr_ref_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    if (p < 0.0) p = 0.0; if (p > 1.0) p = 1.0;
    if (ep < 0.0) ep = 0.0; if (ep > 1.0) ep = 1.0;
    double num = 1.0 - p; if (num < DBL_MIN) num = DBL_MIN;
    double w = 1.0 - ep, pw = 1.0;
    for (int i = 0; i < model_points; ++i) pw = pw * w;
    double denom = 1.0 - pw;
    if (denom < DBL_MIN) return 0;
    num = mfr_ref_det_log(num);
    denom = mfr_ref_det_log(denom);
    if (denom >= 0.0 || -num >= (double)max_iters * (-denom)) return max_iters;
    return (int)rint(num / denom);
}

/* ------------------------------------------------------------------------ */
/* All real roots of a polynomial (ascending coefficients c[0..deg]),        */
/* ascending order.  Derivative-isolation + safeguarded Newton: deterministic*/
/* and libm-free.  deg <= 10.                                                */
/* ------------------------------------------------------------------------ */
#define MAXDEG 10
static double poly_eval(const double *c, int deg, double x)
{
    double y = c[deg];
    for (int i = deg - 1; i >= 0; --i) y = y * x + c[i];
    return y;
}

static double refine_root(const double *c, const double *dc, int deg, double lo, double hi, double flo)
{
    /* safeguarded Newton (Numerical Recipes rtsafe): bisect whenever the Newton step would leave
     * the bracket or is not shrinking it at least as fast as bisection would */
    double x = 0.5 * (lo + hi), dxold = hi - lo, dx = dxold;
    double fx = poly_eval(c, deg, x), dfx = poly_eval(dc, deg - 1, x);
    for (int it = 0; it < 200; ++it) {
        if (fx == 0.0) break;
        if ((fx < 0.0) == (flo < 0.0)) lo = x; else hi = x;
        double a = (x - hi) * dfx - fx, b = (x - lo) * dfx - fx;
        double tf = 2.0 * fx; if (tf < 0.0) tf = -tf;
        double td = dxold * dfx; if (td < 0.0) td = -td;
        int newton = ((a < 0.0) != (b < 0.0)) && (tf <= td);
        double xn;
        dxold = dx;
        if (newton) { dx = fx / dfx; xn = x - dx; }
        else { dx = 0.5 * (hi - lo); xn = lo + dx; }
        if (!(xn > lo && xn < hi)) { dx = 0.5 * (hi - lo); xn = lo + dx; }
        if (xn == x) break;
        double adx = dx < 0.0 ? -dx : dx, ax = xn < 0.0 ? -xn : xn;
        x = xn;
        if (adx <= 2e-16 * ax || adx < 1e-300) break;
        fx = poly_eval(c, deg, x); dfx = poly_eval(dc, deg - 1, x);
    }
    return x;
}

int mfr_ref_poly_real_roots(const double *c_in, int deg, double *roots)
{
    double d[MAXDEG + 1][MAXDEG + 1];
    while (deg > 0 && c_in[deg] == 0.0) --deg;
    if (deg <= 0) return 0;
    double bound = 0.0;
    for (int i = 0; i < deg; ++i) {
        double r = c_in[i] / c_in[deg]; if (r < 0.0) r = -r;
        if (r > bound) bound = r;
    }
    bound = bound + 1.0;
    if (!(bound < 1e300)) return 0;
    for (int i = 0; i <= deg; ++i) d[0][i] = c_in[i];
    for (int L = 1; L < deg; ++L)
        for (int i = 0; i <= deg - L; ++i) d[L][i] = d[L - 1][i + 1] * (double)(i + 1);
    double crit[MAXDEG + 2], cur[MAXDEG + 2];
    int nc = 0;
    {   /* level deg-1 is linear */
        const double *p = d[deg - 1];
        crit[0] = -p[0] / p[1]; nc = 1;
    }
    for (int L = deg - 2; L >= 0; --L) {
        const double *p = d[L]; const double *dp = d[L + 1];
        int m = deg - L, nr = 0;
        double xl = -bound, fl = poly_eval(p, m, xl);
        for (int i = 0; i <= nc; ++i) {
            double xh = (i < nc) ? crit[i] : bound;
            if (i < nc && !(xh > xl)) continue;          /* coincident critical points */
            double fh = poly_eval(p, m, xh);
            if (fl == 0.0) {
                if (nr == 0 || cur[nr - 1] != xl) cur[nr++] = xl;
            } else if (fh != 0.0 && ((fl < 0.0) != (fh < 0.0))) {
                cur[nr++] = refine_root(p, dp, m, xl, xh, fl);
            }
            xl = xh; fl = fh;
        }
        if (fl == 0.0 && (nr == 0 || cur[nr - 1] != xl)) cur[nr++] = xl;
        nc = nr;
        for (int i = 0; i < nr; ++i) crit[i] = cur[i];
        if (nc == 0 && L > 0) {
            /* no critical points: polynomial monotone -> handled naturally by the
             * single bracket (-bound, bound) at the next level */
        }
    }
    for (int i = 0; i < nc; ++i) roots[i] = crit[i];
    return nc;
}

/* ------------------------------------------------------------------------ */
/* pose_solver.py:6-17 backproject_3d                                        */
/* ------------------------------------------------------------------------ */
/* np.linalg.inv on the pinhole matrix [[fx,0,cx],[0,fy,cy],[0,0,1]] in K's own dtype (zero skew / unit bottom row only).
 * The off-diagonal entries are what the LAPACK gesv back-substitution of the numpy build in the reference container gives,
 * pinned by the reference-executed fixtures (oracle/gen_golden.py, 5000 random K each, zero mismatches):
 *   float32 (sgesv): -(cx / fx)          -- a division
 *   float64 (dgesv): -(cx * (1.0 / fx))  -- the triangular solve multiplies by the reciprocal of the pivot */
int mfr_ref_load_intr(const void *K, int k_dtype, mfr_intr *o)
{
    if (k_dtype == MFR_K_F32) {
        const float *k = (const float *)K;
        if (k[1] != 0.f || k[3] != 0.f || k[6] != 0.f || k[7] != 0.f || k[8] != 1.f) return -1;
        if (k[0] == 0.f || k[4] == 0.f) return -1;
        const float ifx = 1.0f / k[0], ify = 1.0f / k[4];
        const float icx = -(k[2] / k[0]), icy = -(k[5] / k[4]);
        o->fx = (double)k[0]; o->fy = (double)k[4]; o->cx = (double)k[2]; o->cy = (double)k[5];
        o->ifx = (double)ifx; o->ify = (double)ify; o->icx = (double)icx; o->icy = (double)icy;
        o->f32 = 1;
        return 0;
    }
    if (k_dtype != MFR_K_F64) return -1;
    const double *k = (const double *)K;
    if (k[1] != 0.0 || k[3] != 0.0 || k[6] != 0.0 || k[7] != 0.0 || k[8] != 1.0) return -1;
    if (k[0] == 0.0 || k[4] == 0.0) return -1;
    o->fx = k[0]; o->fy = k[4]; o->cx = k[2]; o->cy = k[5];
    o->ifx = 1.0 / k[0]; o->ify = 1.0 / k[4];
    o->icx = -(k[2] * o->ifx); o->icy = -(k[5] * o->ify);
    o->f32 = 0;
    return 0;
}

static inline void backproject_intr(const mfr_intr *ki, int32_t ui, int32_t vi, float depth, double *xyz)
{
    /* (inv(K) @ [u, v, 1]) as the f64 matrix product evaluates it (zero entries included), then depth * ray */
    const double u = (double)ui, v = (double)vi, d = (double)depth;
    const double rx = (ki->ifx * u + 0.0 * v) + ki->icx;
    const double ry = (0.0 * u + ki->ify * v) + ki->icy;
    const double rz = (0.0 * u + 0.0 * v) + 1.0;
    xyz[0] = d * rx; xyz[1] = d * ry; xyz[2] = d * rz;
}

int mfr_ref_backproject(const int32_t *uv, const float *depth, int n, const void *K, int k_dtype, double *xyz)
{
    mfr_intr ki;
    if (mfr_ref_load_intr(K, k_dtype, &ki)) return -1;
    for (int i = 0; i < n; ++i) backproject_intr(&ki, uv[2 * i], uv[2 * i + 1], depth[i], xyz + 3 * i);
    return 0;
}

float mfr_ref_depth_min(const float *depth, int hw)
{
    float m = depth[0];
    for (int i = 1; i < hw; ++i) if (depth[i] < m) m = depth[i];
    return m;
}

/* np.int32(x) on float32: C truncation toward zero */
static inline int32_t trunc_i32(float x) { return (int32_t)x; }

/* pose_solver.py:186-206.  Out-of-image pixels (numpy would raise IndexError,
 * or wrap for negatives) are treated as invalid -- documented deviation. */
int mfr_ref_pnp_lift(const float *pts0, const float *pts1, int n, const float *depth0, int H, int W,
                     const void *K0, int k_dtype, double *xyz, double *obs, int32_t *src_idx)
{
    float dmin = mfr_ref_depth_min(depth0, H * W);
    mfr_intr ki;
    if (mfr_ref_load_intr(K0, k_dtype, &ki)) return -1;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        int32_t u = trunc_i32(pts0[2 * i]), v = trunc_i32(pts0[2 * i + 1]);
        if (u < 0 || u >= W || v < 0 || v >= H) continue;
        float d = depth0[v * W + u];
        if (!(d > dmin)) continue;
        backproject_intr(&ki, u, v, d, xyz + 3 * m);
        obs[2 * m] = (double)pts1[2 * i]; obs[2 * m + 1] = (double)pts1[2 * i + 1];
        src_idx[m] = i;
        ++m;
    }
    return m;
}

/* ------------------------------------------------------------------------ */
/* small linear algebra                                                      */
/* ------------------------------------------------------------------------ */
static inline double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void rot_apply(const double *R, const double *t, const double *X, double *Y)
{
    Y[0] = ((R[0] * X[0] + R[1] * X[1]) + R[2] * X[2]) + t[0];
    Y[1] = ((R[3] * X[0] + R[4] * X[1]) + R[5] * X[2]) + t[1];
    Y[2] = ((R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]) + t[2];
}

/* reprojection error^2 in pixels; mirrors cv::projectPoints (z==0 -> 1/z := 1,
 * no cheirality test) as used by PnPRansacCallback::computeError. */
static inline double reproj_err2(const double *R, const double *t, const double *X, const double *x,
                                 const double *Kd)
{
    double Y[3]; rot_apply(R, t, X, Y);
    double iz = (Y[2] != 0.0) ? 1.0 / Y[2] : 1.0;
    double du = (Kd[0] * (Y[0] * iz) + Kd[2]) - x[0];
    double dv = (Kd[1] * (Y[1] * iz) + Kd[3]) - x[1];
    return du * du + dv * dv;
}

/* ------------------------------------------------------------------------ */
/* P3P (Grunert 1841 formulation, quartic in v = s3/s1, built by polynomial  */
/* arithmetic; roots by mfr_ref_poly_real_roots).  X: 3 world points (rows), */
/* f: 3 unit bearing vectors (rows).  Returns up to 4 (R,t): Xc = R X + t.   */
/* Stands in for cv::p3p (Gao et al.) inside solvePnPRansac(SOLVEPNP_P3P).   */
/* ------------------------------------------------------------------------ */
static int frame_from_triangle(const double *P0, const double *P1, const double *P2, double *E /*3x3 cols e1,e2,e3*/)
{
    double a[3] = { P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2] };
    double b[3] = { P2[0] - P0[0], P2[1] - P0[1], P2[2] - P0[2] };
    double na = sqrt(dot3(a, a));
    if (!(na > 0.0)) return -1;
    double e1[3] = { a[0] / na, a[1] / na, a[2] / na };
    double c[3]; cross3(e1, b, c);
    double nc = sqrt(dot3(c, c));
    if (!(nc > 0.0)) return -1;
    double e3[3] = { c[0] / nc, c[1] / nc, c[2] / nc };
    double e2[3]; cross3(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { E[3 * i] = e1[i]; E[3 * i + 1] = e2[i]; E[3 * i + 2] = e3[i]; }
    return 0;
}

int mfr_ref_p3p(const double X[9], const double f[9], double Rs[36], double ts[12])
{
    const double *X0 = X, *X1 = X + 3, *X2 = X + 6;
    const double *f0 = f, *f1 = f + 3, *f2 = f + 6;
    double d12[3] = { X1[0] - X2[0], X1[1] - X2[1], X1[2] - X2[2] };
    double d02[3] = { X0[0] - X2[0], X0[1] - X2[1], X0[2] - X2[2] };
    double d01[3] = { X0[0] - X1[0], X0[1] - X1[1], X0[2] - X1[2] };
    double a2 = dot3(d12, d12), b2 = dot3(d02, d02), c2 = dot3(d01, d01);
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;
    double ca = dot3(f1, f2), cb = dot3(f0, f2), cg = dot3(f0, f1);
    double k1 = (a2 - c2) / b2, q = c2 / b2;
    /* u = N(v)/D(v) */
    double N[3] = { 1.0 + k1, -2.0 * k1 * cb, -1.0 + k1 };
    double D[2] = { 2.0 * cg, -2.0 * ca };
    double S[3] = { 1.0 - q, 2.0 * q * cb, -q };
    /* P = N*N - 2cg*N*D + D*D*S */
    double NN[5] = { N[0] * N[0], 2.0 * (N[0] * N[1]), 2.0 * (N[0] * N[2]) + N[1] * N[1], 2.0 * (N[1] * N[2]), N[2] * N[2] };
    double ND[4] = { N[0] * D[0], N[0] * D[1] + N[1] * D[0], N[1] * D[1] + N[2] * D[0], N[2] * D[1] };
    double DD[3] = { D[0] * D[0], 2.0 * (D[0] * D[1]), D[1] * D[1] };
    double DDS[5] = { DD[0] * S[0], DD[0] * S[1] + DD[1] * S[0], (DD[0] * S[2] + DD[1] * S[1]) + DD[2] * S[0],
                      DD[1] * S[2] + DD[2] * S[1], DD[2] * S[2] };
    double P[5];
    double m2cg = -2.0 * cg;
    P[0] = (NN[0] + m2cg * ND[0]) + DDS[0];
    P[1] = (NN[1] + m2cg * ND[1]) + DDS[1];
    P[2] = (NN[2] + m2cg * ND[2]) + DDS[2];
    P[3] = (NN[3] + m2cg * ND[3]) + DDS[3];
    P[4] = NN[4] + DDS[4];
    double roots[4];
    int nr = mfr_ref_poly_real_roots(P, 4, roots);
    double EW[9];
    if (frame_from_triangle(X0, X1, X2, EW)) return 0;
    int ns = 0;
    for (int r = 0; r < nr; ++r) {
        double v = roots[r];
        if (!(v > 0.0)) continue;
        double Dv = D[1] * v + D[0];
        if (Dv == 0.0) continue;
        double u = ((N[2] * v + N[1]) * v + N[0]) / Dv;
        if (!(u > 0.0)) continue;
        double den = (1.0 + v * v) - 2.0 * v * cb;
        if (!(den > 0.0)) continue;
        double s0 = sqrt(b2 / den), s1 = u * s0, s2 = v * s0;
        double P0[3] = { s0 * f0[0], s0 * f0[1], s0 * f0[2] };
        double P1[3] = { s1 * f1[0], s1 * f1[1], s1 * f1[2] };
        double P2[3] = { s2 * f2[0], s2 * f2[1], s2 * f2[2] };
        double EC[9];
        if (frame_from_triangle(P0, P1, P2, EC)) continue;
        double *R = Rs + 9 * ns, *t = ts + 3 * ns;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                R[3 * i + j] = (EC[3 * i] * EW[3 * j] + EC[3 * i + 1] * EW[3 * j + 1]) + EC[3 * i + 2] * EW[3 * j + 2];
        for (int i = 0; i < 3; ++i)
            t[i] = P0[i] - ((R[3 * i] * X0[0] + R[3 * i + 1] * X0[1]) + R[3 * i + 2] * X0[2]);
        ++ns;
    }
    return ns;
}

/* one RANSAC hypothesis: P3P on sample[0..2], disambiguated by sample[3]
 * (cv::solvePnP(P3P) with 4 points keeps the solution with the smallest
 * reprojection error on the 4th).  Returns 1 if a model was produced. */
static int pnp_hypothesis(const double *xyz, const double *obs, const int *s, const double *Kd,
                          double *R, double *t)
{
    double X[9], f[9];
    for (int k = 0; k < 3; ++k) {
        const double *p = xyz + 3 * s[k]; const double *o = obs + 2 * s[k];
        X[3 * k] = p[0]; X[3 * k + 1] = p[1]; X[3 * k + 2] = p[2];
        double bx = (o[0] - Kd[2]) / Kd[0], by = (o[1] - Kd[3]) / Kd[1];
        double nn = sqrt((bx * bx + by * by) + 1.0);
        f[3 * k] = bx / nn; f[3 * k + 1] = by / nn; f[3 * k + 2] = 1.0 / nn;
    }
    double Rs[36], ts[12];
    int ns = mfr_ref_p3p(X, f, Rs, ts);
    if (ns <= 0) return 0;
    int best = -1; double beste = 0.0;
    for (int i = 0; i < ns; ++i) {
        double e = reproj_err2(Rs + 9 * i, ts + 3 * i, xyz + 3 * s[3], obs + 2 * s[3], Kd);
        if (!(e == e)) continue;
        if (best < 0 || e < beste) { best = i; beste = e; }
    }
    if (best < 0) return 0;
    memcpy(R, Rs + 9 * best, 72); memcpy(t, ts + 3 * best, 24);
    return 1;
}

/* ------------------------------------------------------------------------ */
/* wave64-ordered reduction (matches the device's lane-strided partials +    */
/* xor butterfly).  vals: n_terms x nacc, term i goes to lane (i & 63).       */
/* ------------------------------------------------------------------------ */
#define NACC 28
typedef struct { double a[64][NACC]; } wave_acc_t;
static void wave_acc_zero(wave_acc_t *w) { memset(w, 0, sizeof(*w)); }
static void wave_acc_finish(wave_acc_t *w, int nacc, double *out)
{
    for (int off = 32; off >= 1; off >>= 1) {
        double tmp[64][NACC];
        for (int l = 0; l < 64; ++l)
            for (int k = 0; k < nacc; ++k) tmp[l][k] = w->a[l][k] + w->a[l ^ off][k];
        memcpy(w->a, tmp, sizeof(tmp));
    }
    for (int k = 0; k < nacc; ++k) out[k] = w->a[0][k];
}

static void quat_right_update(const double *R, const double *dw, double *Rn)
{
    /* Rn = R * Rot(q), q = normalise(1, dw/2) */
    double hx = 0.5 * dw[0], hy = 0.5 * dw[1], hz = 0.5 * dw[2];
    double nn = sqrt(((hx * hx + hy * hy) + hz * hz) + 1.0);
    double w = 1.0 / nn, x = hx / nn, y = hy / nn, z = hz / nn;
    double Q[9];
    Q[0] = 1.0 - 2.0 * (y * y + z * z); Q[1] = 2.0 * (x * y - w * z);       Q[2] = 2.0 * (x * z + w * y);
    Q[3] = 2.0 * (x * y + w * z);       Q[4] = 1.0 - 2.0 * (x * x + z * z); Q[5] = 2.0 * (y * z - w * x);
    Q[6] = 2.0 * (x * z - w * y);       Q[7] = 2.0 * (y * z + w * x);       Q[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = (R[3 * i] * Q[j] + R[3 * i + 1] * Q[3 + j]) + R[3 * i + 2] * Q[6 + j];
}

static double pnp_cost(const double *xyz, const double *obs, const int32_t *idx, int n, const double *Kd,
                       const double *R, const double *t)
{
    wave_acc_t *w = (wave_acc_t *)malloc(sizeof(wave_acc_t));
    wave_acc_zero(w);
    for (int i = 0; i < n; ++i) {
        int j = idx ? idx[i] : i;
        w->a[i & 63][0] = w->a[i & 63][0] + reproj_err2(R, t, xyz + 3 * j, obs + 2 * j, Kd);
    }
    double c; wave_acc_finish(w, 1, &c);
    free(w);
    return c;
}

/* solve 6x6 SPD system by Cholesky; returns 0 on success */
static int chol_solve6(const double *A /*36 row-major sym*/, const double *b, double *x)
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = A[6 * i + j];
            for (int k = 0; k < j; ++k) s = s - L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0.0)) return -1;
                L[6 * i + i] = sqrt(s);
            } else {
                L[6 * i + j] = s / L[6 * j + j];
            }
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s = s - L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s = s - L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return 0;
}

/* Levenberg-Marquardt on the reprojection error over idx[0..n_idx) -- stands in
 * for (a) the EPnP refit inside cv::solvePnPRansac and (b) the reference's
 * cv.solvePnPGeneric(SOLVEPNP_ITERATIVE, useExtrinsicGuess) (pose_solver.py:216-220).
 * Rotation update is a right-multiplied unit quaternion (no trig). */
int mfr_ref_pnp_lm(const double *xyz, const double *obs, const int32_t *idx, int n_idx,
                   const double Kd[4], int max_iter, double R[9], double t[3])
{
    double lambda = 1e-3;
    double cost = pnp_cost(xyz, obs, idx, n_idx, Kd, R, t);
    if (!(cost == cost) || !(cost < 1e300)) return -1;
    wave_acc_t *w = (wave_acc_t *)malloc(sizeof(wave_acc_t));
    for (int it = 0; it < max_iter; ++it) {
        wave_acc_zero(w);
        for (int i = 0; i < n_idx; ++i) {
            int j = idx ? idx[i] : i;
            const double *X = xyz + 3 * j, *x = obs + 2 * j;
            double Y[3]; rot_apply(R, t, X, Y);
            double iz = (Y[2] != 0.0) ? 1.0 / Y[2] : 1.0;
            double xn = Y[0] * iz, yn = Y[1] * iz;
            double ru = (Kd[0] * xn + Kd[2]) - x[0];
            double rv = (Kd[1] * yn + Kd[3]) - x[1];
            /* d(Xc)/d(dw) columns: R * (e_k x X) */
            double a0[3] = { 0.0, -X[2], X[1] }, a1[3] = { X[2], 0.0, -X[0] }, a2[3] = { -X[1], X[0], 0.0 };
            double G[3][6];
            for (int r = 0; r < 3; ++r) {
                G[r][0] = (R[3 * r] * a0[0] + R[3 * r + 1] * a0[1]) + R[3 * r + 2] * a0[2];
                G[r][1] = (R[3 * r] * a1[0] + R[3 * r + 1] * a1[1]) + R[3 * r + 2] * a1[2];
                G[r][2] = (R[3 * r] * a2[0] + R[3 * r + 1] * a2[1]) + R[3 * r + 2] * a2[2];
                G[r][3] = (r == 0) ? 1.0 : 0.0; G[r][4] = (r == 1) ? 1.0 : 0.0; G[r][5] = (r == 2) ? 1.0 : 0.0;
            }
            double pu0 = Kd[0] * iz, pu2 = -(Kd[0] * xn) * iz;
            double pv1 = Kd[1] * iz, pv2 = -(Kd[1] * yn) * iz;
            double Ju[6], Jv[6];
            for (int k = 0; k < 6; ++k) {
                Ju[k] = pu0 * G[0][k] + pu2 * G[2][k];
                Jv[k] = pv1 * G[1][k] + pv2 * G[2][k];
            }
            double *acc = w->a[i & 63];
            int q = 0;
            for (int r = 0; r < 6; ++r)
                for (int c = r; c < 6; ++c, ++q)
                    acc[q] = acc[q] + (Ju[r] * Ju[c] + Jv[r] * Jv[c]);
            for (int r = 0; r < 6; ++r, ++q)
                acc[q] = acc[q] + (Ju[r] * ru + Jv[r] * rv);
        }
        double s[27]; wave_acc_finish(w, 27, s);
        double H[36], g[6];
        { int q = 0;
          for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c, ++q) { H[6 * r + c] = s[q]; H[6 * c + r] = s[q]; }
          for (int r = 0; r < 6; ++r, ++q) g[r] = -s[q]; }
        for (int r = 0; r < 6; ++r) H[6 * r + r] = H[6 * r + r] + lambda * H[6 * r + r];
        double dlt[6];
        if (chol_solve6(H, g, dlt)) { lambda = lambda * 10.0; if (lambda > 1e12) break; continue; }
        double Rn[9], tn[3];
        quat_right_update(R, dlt, Rn);
        tn[0] = t[0] + dlt[3]; tn[1] = t[1] + dlt[4]; tn[2] = t[2] + dlt[5];
        double cn = pnp_cost(xyz, obs, idx, n_idx, Kd, Rn, tn);
        double mx = 0.0;
        for (int k = 0; k < 6; ++k) { double a = dlt[k] < 0.0 ? -dlt[k] : dlt[k]; if (a > mx) mx = a; }
        if (cn < cost) {
            double dec = cost - cn;
            memcpy(R, Rn, 72); memcpy(t, tn, 24);
            int done = (dec <= 1e-14 * cost);
            cost = cn;
            lambda = lambda * 0.1; if (lambda < 1e-12) lambda = 1e-12;
            if (done) break;
        } else {
            lambda = lambda * 10.0; if (lambda > 1e12) break;
        }
        if (mx < 1e-13) break;
    }
    free(w);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* cv.solvePnPRansac(flags=SOLVEPNP_P3P) restatement (pose_solver.py:209-213) */
/* = RANSACPointSetRegistrator::run (ptsetreg.cpp): model size 4, inlier test*/
/* err^2 <= thr^2, best = strictly more inliers (and > 3), adaptive iteration */
/* cap.  All max_iters hypotheses are a pure function of (seed,pair,iter), so */
/* the device evaluates them in parallel and replays this loop as a scan.    */
/* ------------------------------------------------------------------------ */
int mfr_ref_pnp_ransac(const double *xyz, const double *obs, int n, const void *K1, int k_dtype,
                       int max_iters, double thr, double conf, uint64_t seed, uint64_t pair_id,
                       double R[9], double t[3], uint8_t *mask, int *n_inl,
                       int *best_iter, int *iters_run, int32_t *counts)
{
    /* K1.numpy() handed to OpenCV, which works in double: exact widening for float32 (pose_solver.py:209-213) */
    mfr_intr k1;
    if (mfr_ref_load_intr(K1, k_dtype, &k1)) return MFR_ST_NO_MODEL;
    const double Kd[4] = { k1.fx, k1.fy, k1.cx, k1.cy };
    const double thr2 = thr * thr;
    for (int i = 0; i < 9; ++i) R[i] = NAN;
    for (int i = 0; i < 3; ++i) t[i] = NAN;
    *n_inl = 0; if (best_iter) *best_iter = -1; if (iters_run) *iters_run = 0;
    if (mask) memset(mask, 0, (size_t)n);
    if (n < 4) return MFR_ST_TOO_FEW;
    if (max_iters < 1) max_iters = 1;
    double bR[9], bt[3]; int best = 3, bit = -1;
    int niters = max_iters, it;
    if (n == 4) {
        int s[4] = { 0, 1, 2, 3 };
        if (!pnp_hypothesis(xyz, obs, s, Kd, bR, bt)) return MFR_ST_NO_MODEL;
        best = 4; bit = 0; it = 1;
        if (mask) memset(mask, 1, 4);
    } else {
        for (it = 0; it < niters; ++it) {
            int s[4]; double hR[9], ht[3];
            mfr_ref_sample_distinct(seed, pair_id, (uint32_t)it, n, 4, s);
            int cnt = 0;
            if (pnp_hypothesis(xyz, obs, s, Kd, hR, ht))
                for (int i = 0; i < n; ++i)
                    cnt += (reproj_err2(hR, ht, xyz + 3 * i, obs + 2 * i, Kd) <= thr2);
            if (counts) counts[it] = cnt;
            if (cnt > best) {
                best = cnt; bit = it;
                memcpy(bR, hR, 72); memcpy(bt, ht, 24);
                niters = mfr_ref_update_num_iters(conf, (double)(n - cnt) / (double)n, 4, niters);
            }
        }
        if (counts) for (int k = it; k < max_iters; ++k) counts[k] = -1;
        if (bit < 0) { if (iters_run) *iters_run = it; return MFR_ST_NO_MODEL; }
    }
    if (iters_run) *iters_run = it;
    if (best_iter) *best_iter = bit;
    /* inlier list of the best model */
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        int in = (n == 4) ? 1 : (reproj_err2(bR, bt, xyz + 3 * i, obs + 2 * i, Kd) <= thr2);
        if (mask) mask[i] = (uint8_t)in;
        if (in) idx[m++] = i;
    }
    *n_inl = m;
    int st = MFR_ST_OK;
    if (n > 4) {
        /* non-minimal refit on the inliers (OpenCV: EPnP; here LM from the best
         * minimal model -- documented substitution), then the reference's
         * ITERATIVE refinement when >= 6 inliers (pose_solver.py:216-220). */
        if (mfr_ref_pnp_lm(xyz, obs, idx, m, Kd, 20, bR, bt)) st = MFR_ST_NO_MODEL;
        if (st == MFR_ST_OK && m >= 6)
            if (mfr_ref_pnp_lm(xyz, obs, idx, m, Kd, 20, bR, bt)) st = MFR_ST_NO_MODEL;
    }
    free(idx);
    if (st == MFR_ST_OK) {
        for (int i = 0; i < 9; ++i) if (!(bR[i] == bR[i])) st = MFR_ST_NO_MODEL;
        for (int i = 0; i < 3; ++i) if (!(bt[i] == bt[i])) st = MFR_ST_NO_MODEL;
    }
    if (st == MFR_ST_OK) {
        double tn = sqrt(dot3(bt, bt));
        if (tn > 1000.0) st = MFR_ST_DEGENERATE;     /* pose_solver.py:223-225 */
    }
    if (st == MFR_ST_OK) { memcpy(R, bR, 72); memcpy(t, bt, 24); }
    else { *n_inl = 0; }
    return st;
}

int mfr_ref_pnp_solve(const float *pts0, const float *pts1, int n, const float *depth0, int H, int W,
                      const void *K0, const void *K1, int k_dtype, int max_iters, double thr, double conf,
                      uint64_t seed, uint64_t pair_id, double R[9], double t[3], int *n_inl)
{
    for (int i = 0; i < 9; ++i) R[i] = NAN;
    for (int i = 0; i < 3; ++i) t[i] = NAN;
    *n_inl = 0;
    if (n < 4) return MFR_ST_TOO_FEW;                  /* pose_solver.py:188-189 */
    double *xyz = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    double *obs = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    int32_t *src = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int m = mfr_ref_pnp_lift(pts0, pts1, n, depth0, H, W, K0, k_dtype, xyz, obs, src);
    int st;
    if (m < 4) st = MFR_ST_BAD_DEPTH;                  /* pose_solver.py:197-198 */
    else st = mfr_ref_pnp_ransac(xyz, obs, m, K1, k_dtype, max_iters, thr, conf, seed, pair_id, R, t, NULL, n_inl,
                                 NULL, NULL, NULL);
    free(xyz); free(obs); free(src);
    return st;
}

/* ------------------------------------------------------------------------ */
/* pose_solver.py:137-172  EssentialMatrixMetricSolver: scale from depth     */
/* ------------------------------------------------------------------------ */
int mfr_ref_scale_lift(const float *pts0, const float *pts1, const uint8_t *mask, int n,
                       const float *depth0, const float *depth1, int H, int W,
                       const void *K0, const void *K1, int k_dtype, const double R[9], const double t[3],
                       double *scale)
{
    int m = 0;
    mfr_intr ki0, ki1;
    if (mfr_ref_load_intr(K0, k_dtype, &ki0) || mfr_ref_load_intr(K1, k_dtype, &ki1)) return -1;
    for (int i = 0; i < n; ++i) {
        if (mask && mask[i] != 1) continue;                                 /* :137 mask == 1 */
        int32_t u0 = trunc_i32(pts0[2 * i]), v0 = trunc_i32(pts0[2 * i + 1]);   /* :138 */
        int32_t u1 = trunc_i32(pts1[2 * i]), v1 = trunc_i32(pts1[2 * i + 1]);   /* :139 */
        if (u0 < 0 || u0 >= W || v0 < 0 || v0 >= H || u1 < 0 || u1 >= W || v1 < 0 || v1 >= H) continue;
        float d0 = depth0[v0 * W + u0], d1 = depth1[v1 * W + u1];           /* :140-141 */
        if (!(d0 > 0.f) || !(d1 > 0.f)) continue;                           /* :144 */
        double p0[3], p1[3], rp0[3];
        backproject_intr(&ki0, u0, v0, d0, p0);                             /* :150 */
        backproject_intr(&ki1, u1, v1, d1, p1);                             /* :151 */
        rp0[0] = (R[0] * p0[0] + R[1] * p0[1]) + R[2] * p0[2];              /* :154 */
        rp0[1] = (R[3] * p0[0] + R[4] * p0[1]) + R[5] * p0[2];
        rp0[2] = (R[6] * p0[0] + R[7] * p0[1]) + R[8] * p0[2];
        double d[3] = { p1[0] - rp0[0], p1[1] - rp0[1], p1[2] - rp0[2] };
        scale[m++] = dot3(d, t);                                            /* :157 */
    }
    return m;
}

int mfr_ref_scale_ransac(const double *scale, int n, double thr, double *best_scale, int *best_idx)
{
    int best = 0, bi = -1;                                                   /* :160-166 */
    for (int i = 0; i < n; ++i) {
        int c = 0;
        for (int j = 0; j < n; ++j) {
            double d = scale[j] - scale[i]; if (d < 0.0) d = -d;
            c += (d < thr);
        }
        if (c > best) { best = c; bi = i; }
    }
    if (best_idx) *best_idx = bi;
    if (best_scale) *best_scale = (bi >= 0) ? scale[bi] : NAN;
    return best;
}
