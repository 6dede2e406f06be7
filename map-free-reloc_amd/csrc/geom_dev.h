// geom_dev.h -- per-lane (scalar) device geometry for the RANSAC kernels (gfx950).
//
// Every function here is the device statement of a step of the pose-solver leg of the
// reference (lib/models/matching/pose_solver.py) or of the OpenCV routine that leg calls.
// Floating-point contract (what makes inlier sets reproducible bit-for-bit against the CPU
// oracle): IEEE binary64, only + - * / sqrt and comparisons, NO fma contraction (this TU is
// compiled with -ffp-contract=off), fixed evaluation order (parenthesised below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MFR_DEV static __device__ __forceinline__
#define MFR_DEV_NOINLINE static __device__ __noinline__

namespace mfr {

// intrinsics dtype tags (values of include/mfr_hip.h MFR_K_F32 / MFR_K_F64; this header is also used without it)
#ifndef MFR_K_F32
#define MFR_K_F32 0
#define MFR_K_F64 1
#endif
static inline bool k_dtype_ok(int k_dtype) { return k_dtype == MFR_K_F32 || k_dtype == MFR_K_F64; }

// ---------------------------------------------------------------- Philox4x32-10
MFR_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// k (<=8) distinct indices in [0,n), draw order preserved.  Replaces OpenCV's
// RANSACPointSetRegistrator::getSubset: counter-based, so hypothesis `iter` of pair
// `pair_id` is a pure function of (seed, pair_id, iter) and all hypotheses can be
// evaluated concurrently.
template <int K>
MFR_DEV void sample_distinct(uint64_t seed, uint64_t pair_id, uint32_t iter, int n, int out[K])
{
    int sorted[K];
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if ((j & 3) == 0)
            philox4x32_10(iter, (uint32_t)(j >> 2), (uint32_t)pair_id, (uint32_t)(pair_id >> 32),
                          (uint32_t)seed, (uint32_t)(seed >> 32), w);
        int v = (int)__umulhi(w[j & 3], (uint32_t)(n - j));
        int pos = 0;
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (q < j && q == pos && v >= sorted[q]) { ++v; ++pos; }
#pragma unroll
        for (int q = K - 1; q > 0; --q)
            if (q <= j && q > pos) sorted[q] = sorted[q - 1];
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (q == pos) sorted[q] = v;
        out[j] = v;
    }
}

// ---------------------------------------------------------------- libm-free log
MFR_DEV double det_log(double x)
{
    uint64_t b = (uint64_t)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) {
        x = x * 18014398509481984.0;
        b = (uint64_t)__double_as_longlong(x);
        e = (int)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)b);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double acc = 0.0;
    for (int k = 17; k >= 0; --k) acc = acc * s2 + 1.0 / (double)(2 * k + 1);
    return 2.0 * s * acc + (double)e * 0.6931471805599453;
}

// OpenCV RANSACUpdateNumIters (calib3d ptsetreg.cpp)
MFR_DEV int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    if (p < 0.0) p = 0.0;
    if (p > 1.0) p = 1.0;
    if (ep < 0.0) ep = 0.0;
    if (ep > 1.0) ep = 1.0;
    double num = 1.0 - p;
    if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
    const double w = 1.0 - ep;
    double pw = 1.0;
    for (int i = 0; i < model_points; ++i) pw = pw * w;
    double denom = 1.0 - pw;
    if (denom < 2.2250738585072014e-308) return 0;
    num = det_log(num);
    denom = det_log(denom);
    if (denom >= 0.0 || -num >= (double)max_iters * (-denom)) return max_iters;
    return (int)__builtin_rint(num / denom);
}

// ---------------------------------------------------------------- polynomial real roots
template <int MAXDEG>
MFR_DEV double poly_eval(const double *c, int deg, double x)
{
    double y = c[deg];
    for (int i = deg - 1; i >= 0; --i) y = y * x + c[i];
    return y;
}

template <int MAXDEG>
MFR_DEV double refine_root(const double *c, const double *dc, int deg, double lo, double hi, double flo)
{
    // safeguarded Newton (rtsafe): bisect whenever the Newton step would leave the bracket or is
    // not shrinking it at least as fast as bisection would
    double x = 0.5 * (lo + hi), dxold = hi - lo, dx = dxold;
    double fx = poly_eval<MAXDEG>(c, deg, x), dfx = poly_eval<MAXDEG>(dc, deg - 1, x);
    for (int it = 0; it < 200; ++it) {
        if (fx == 0.0) break;
        if ((fx < 0.0) == (flo < 0.0)) lo = x; else hi = x;
        const double a = (x - hi) * dfx - fx, b = (x - lo) * dfx - fx;
        double tf = 2.0 * fx;
        if (tf < 0.0) tf = -tf;
        double td = dxold * dfx;
        if (td < 0.0) td = -td;
        const bool newton = ((a < 0.0) != (b < 0.0)) && (tf <= td);
        double xn;
        dxold = dx;
        if (newton) { dx = fx / dfx; xn = x - dx; }
        else { dx = 0.5 * (hi - lo); xn = lo + dx; }
        if (!(xn > lo && xn < hi)) { dx = 0.5 * (hi - lo); xn = lo + dx; }
        if (xn == x) break;
        const double adx = dx < 0.0 ? -dx : dx, ax = xn < 0.0 ? -xn : xn;
        x = xn;
        if (adx <= 2e-16 * ax || adx < 1e-300) break;
        fx = poly_eval<MAXDEG>(c, deg, x);
        dfx = poly_eval<MAXDEG>(dc, deg - 1, x);
    }
    return x;
}

// All real roots (ascending) of c[0..deg]; derivative isolation + safeguarded Newton.
template <int MAXDEG>
MFR_DEV_NOINLINE int poly_real_roots(const double *c_in, int deg, double *roots)
{
    double d[MAXDEG][MAXDEG + 1];
    while (deg > 0 && c_in[deg] == 0.0) --deg;
    if (deg <= 0) return 0;
    double bound = 0.0;
    for (int i = 0; i < deg; ++i) {
        double r = c_in[i] / c_in[deg];
        if (r < 0.0) r = -r;
        if (r > bound) bound = r;
    }
    bound = bound + 1.0;
    if (!(bound < 1e300)) return 0;
    for (int i = 0; i <= deg; ++i) d[0][i] = c_in[i];
    for (int L = 1; L < deg; ++L)
        for (int i = 0; i <= deg - L; ++i) d[L][i] = d[L - 1][i + 1] * (double)(i + 1);
    double crit[MAXDEG + 2], cur[MAXDEG + 2];
    int nc = 1;
    crit[0] = -d[deg - 1][0] / d[deg - 1][1];
    for (int L = deg - 2; L >= 0; --L) {
        const double *p = d[L];
        const double *dp = d[L + 1];
        const int m = deg - L;
        int nr = 0;
        double xl = -bound, fl = poly_eval<MAXDEG>(p, m, xl);
        for (int i = 0; i <= nc; ++i) {
            const double xh = (i < nc) ? crit[i] : bound;
            if (i < nc && !(xh > xl)) continue;
            const double fh = poly_eval<MAXDEG>(p, m, xh);
            if (fl == 0.0) {
                if (nr == 0 || cur[nr - 1] != xl) cur[nr++] = xl;
            } else if (fh != 0.0 && ((fl < 0.0) != (fh < 0.0))) {
                cur[nr++] = refine_root<MAXDEG>(p, dp, m, xl, xh, fl);
            }
            xl = xh; fl = fh;
        }
        if (fl == 0.0 && (nr == 0 || cur[nr - 1] != xl)) cur[nr++] = xl;
        nc = nr;
        for (int i = 0; i < nr; ++i) crit[i] = cur[i];
    }
    for (int i = 0; i < nc; ++i) roots[i] = crit[i];
    return nc;
}

// ---------------------------------------------------------------- small vector helpers
MFR_DEV double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
MFR_DEV void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
MFR_DEV void rot_apply(const double *R, const double *t, const double *X, double *Y)
{
    Y[0] = ((R[0] * X[0] + R[1] * X[1]) + R[2] * X[2]) + t[0];
    Y[1] = ((R[3] * X[0] + R[4] * X[1]) + R[5] * X[2]) + t[1];
    Y[2] = ((R[6] * X[0] + R[7] * X[1]) + R[8] * X[2]) + t[2];
}

// squared reprojection error in pixels; cv::projectPoints semantics (z == 0 -> 1/z := 1, no
// cheirality test) as used by PnPRansacCallback::computeError.  Kd = {fx, fy, cx, cy}.
MFR_DEV double reproj_err2(const double *R, const double *t, const double *X, const double *x, const double *Kd)
{
    double Y[3];
    rot_apply(R, t, X, Y);
    const double iz = (Y[2] != 0.0) ? 1.0 / Y[2] : 1.0;
    const double du = (Kd[0] * (Y[0] * iz) + Kd[2]) - x[0];
    const double dv = (Kd[1] * (Y[1] * iz) + Kd[3]) - x[1];
    return du * du + dv * dv;
}

// Intrinsics cross the C-ABI in the dtype the `data` dict holds them (include/mfr_hip.h MFR_K_F32 / MFR_K_F64); pair b's matrix
// is the b-th block of 9 values.
// kparams: {fx, fy, cx, cy} as stored (float32 widens exactly: what K.numpy() handed to OpenCV becomes, pose_solver.py:209-213).
MFR_DEV void kparams(const void *K, int k_dtype, int b, double Kd[4])
{
    if (k_dtype == MFR_K_F32) {
        const float *k = (const float *)K + 9 * (size_t)b;
        Kd[0] = (double)k[0]; Kd[1] = (double)k[4]; Kd[2] = (double)k[2]; Kd[3] = (double)k[5];
    } else {
        const double *k = (const double *)K + 9 * (size_t)b;
        Kd[0] = k[0]; Kd[1] = k[4]; Kd[2] = k[2]; Kd[3] = k[5];
    }
}
// kinv: np.linalg.inv(K) of the pinhole matrix, evaluated in K's OWN dtype (pose_solver.py:16), widened to double:
// Ki = {inv[0,0], inv[0,2], inv[1,1], inv[1,2]}.  float32 (quirk Q5, resize=None datasets): {1/fx, -(cx/fx), ..} (LAPACK sgesv
// back-substitution divides); float64 (the Map-free loader: lib/datasets/utils.py:117-130): {1/fx, -(cx*(1/fx)), ..} (dgesv's
// triangular solve multiplies by the reciprocal pivot).  Both forms are pinned bit-for-bit by the reference-executed fixtures
// (tests/golden/ref_backproject.npz, ref_k64.npz) through oracle/mfr_oracle.c:mfr_ref_load_intr, which this mirrors.
MFR_DEV void kinv(const void *K, int k_dtype, int b, double Ki[4])
{
    if (k_dtype == MFR_K_F32) {
        const float *k = (const float *)K + 9 * (size_t)b;
        const float ifx = 1.0f / k[0], icx = -(k[2] / k[0]), ify = 1.0f / k[4], icy = -(k[5] / k[4]);
        Ki[0] = (double)ifx; Ki[1] = (double)icx; Ki[2] = (double)ify; Ki[3] = (double)icy;
    } else {
        const double *k = (const double *)K + 9 * (size_t)b;
        Ki[0] = 1.0 / k[0]; Ki[2] = 1.0 / k[4];
        Ki[1] = -(k[2] * Ki[0]); Ki[3] = -(k[5] * Ki[2]);
    }
}
// pose_solver.py:6-17: ray = inv(K) @ [u, v, 1] as the float64 matrix product evaluates it, xyz = depth * ray
MFR_DEV void backproject(int u, int v, float depth, const double Ki[4], double *xyz)
{
    const double du = (double)u, dv = (double)v, d = (double)depth;
    const double rx = (Ki[0] * du + 0.0 * dv) + Ki[1];
    const double ry = (0.0 * du + Ki[2] * dv) + Ki[3];
    const double rz = (0.0 * du + 0.0 * dv) + 1.0;
    xyz[0] = d * rx; xyz[1] = d * ry; xyz[2] = d * rz;
}

// ---------------------------------------------------------------- P3P (Grunert)
MFR_DEV int frame_from_triangle(const double *P0, const double *P1, const double *P2, double *E)
{
    const double a[3] = { P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2] };
    const double b[3] = { P2[0] - P0[0], P2[1] - P0[1], P2[2] - P0[2] };
    const double na = sqrt(dot3(a, a));
    if (!(na > 0.0)) return -1;
    const double e1[3] = { a[0] / na, a[1] / na, a[2] / na };
    double c[3];
    cross3(e1, b, c);
    const double nc = sqrt(dot3(c, c));
    if (!(nc > 0.0)) return -1;
    const double e3[3] = { c[0] / nc, c[1] / nc, c[2] / nc };
    double e2[3];
    cross3(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { E[3 * i] = e1[i]; E[3 * i + 1] = e2[i]; E[3 * i + 2] = e3[i]; }
    return 0;
}

// X: 3 world points (rows), f: 3 unit bearings (rows).  Up to 4 (R,t), Xc = R X + t.
// Stands in for cv::p3p inside solvePnPRansac(SOLVEPNP_P3P) (pose_solver.py:209-213).
MFR_DEV_NOINLINE int p3p(const double *X, const double *f, double *Rs, double *ts)
{
    const double *X0 = X, *X1 = X + 3, *X2 = X + 6;
    const double *f0 = f, *f1 = f + 3, *f2 = f + 6;
    const double d12[3] = { X1[0] - X2[0], X1[1] - X2[1], X1[2] - X2[2] };
    const double d02[3] = { X0[0] - X2[0], X0[1] - X2[1], X0[2] - X2[2] };
    const double d01[3] = { X0[0] - X1[0], X0[1] - X1[1], X0[2] - X1[2] };
    const double a2 = dot3(d12, d12), b2 = dot3(d02, d02), c2 = dot3(d01, d01);
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;
    const double ca = dot3(f1, f2), cb = dot3(f0, f2), cg = dot3(f0, f1);
    const double k1 = (a2 - c2) / b2, q = c2 / b2;
    const double N[3] = { 1.0 + k1, -2.0 * k1 * cb, -1.0 + k1 };
    const double D[2] = { 2.0 * cg, -2.0 * ca };
    const double S[3] = { 1.0 - q, 2.0 * q * cb, -q };
    const double NN[5] = { N[0] * N[0], 2.0 * (N[0] * N[1]), 2.0 * (N[0] * N[2]) + N[1] * N[1],
                           2.0 * (N[1] * N[2]), N[2] * N[2] };
    const double ND[4] = { N[0] * D[0], N[0] * D[1] + N[1] * D[0], N[1] * D[1] + N[2] * D[0], N[2] * D[1] };
    const double DD[3] = { D[0] * D[0], 2.0 * (D[0] * D[1]), D[1] * D[1] };
    const double DDS[5] = { DD[0] * S[0], DD[0] * S[1] + DD[1] * S[0],
                            (DD[0] * S[2] + DD[1] * S[1]) + DD[2] * S[0],
                            DD[1] * S[2] + DD[2] * S[1], DD[2] * S[2] };
    const double m2cg = -2.0 * cg;
    double P[5];
    P[0] = (NN[0] + m2cg * ND[0]) + DDS[0];
    P[1] = (NN[1] + m2cg * ND[1]) + DDS[1];
    P[2] = (NN[2] + m2cg * ND[2]) + DDS[2];
    P[3] = (NN[3] + m2cg * ND[3]) + DDS[3];
    P[4] = NN[4] + DDS[4];
    double roots[4];
    const int nr = poly_real_roots<4>(P, 4, roots);
    double EW[9];
    if (frame_from_triangle(X0, X1, X2, EW)) return 0;
    int ns = 0;
    for (int r = 0; r < nr; ++r) {
        const double v = roots[r];
        if (!(v > 0.0)) continue;
        const double Dv = D[1] * v + D[0];
        if (Dv == 0.0) continue;
        const double u = ((N[2] * v + N[1]) * v + N[0]) / Dv;
        if (!(u > 0.0)) continue;
        const double den = (1.0 + v * v) - 2.0 * v * cb;
        if (!(den > 0.0)) continue;
        const double s0 = sqrt(b2 / den), s1 = u * s0, s2 = v * s0;
        const double P0[3] = { s0 * f0[0], s0 * f0[1], s0 * f0[2] };
        const double P1[3] = { s1 * f1[0], s1 * f1[1], s1 * f1[2] };
        const double P2[3] = { s2 * f2[0], s2 * f2[1], s2 * f2[2] };
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

// One RANSAC hypothesis of cv::solvePnPRansac(P3P): P3P on samples 0..2, disambiguated by the
// reprojection error of sample 3.  xyz [n,3], obs [n,2] are the pair's lifted points.
MFR_DEV int pnp_hypothesis(const double *xyz, const double *obs, const int *s, const double *Kd,
                           double *R, double *t)
{
    double X[9], f[9];
    for (int k = 0; k < 3; ++k) {
        const double *p = xyz + 3 * s[k];
        const double *o = obs + 2 * s[k];
        X[3 * k] = p[0]; X[3 * k + 1] = p[1]; X[3 * k + 2] = p[2];
        const double bx = (o[0] - Kd[2]) / Kd[0], by = (o[1] - Kd[3]) / Kd[1];
        const double nn = sqrt((bx * bx + by * by) + 1.0);
        f[3 * k] = bx / nn; f[3 * k + 1] = by / nn; f[3 * k + 2] = 1.0 / nn;
    }
    double Rs[36], ts[12];
    const int ns = p3p(X, f, Rs, ts);
    if (ns <= 0) return 0;
    int best = -1;
    double beste = 0.0;
    for (int i = 0; i < ns; ++i) {
        const double e = reproj_err2(Rs + 9 * i, ts + 3 * i, xyz + 3 * s[3], obs + 2 * s[3], Kd);
        if (!(e == e)) continue;
        if (best < 0 || e < beste) { best = i; beste = e; }
    }
    if (best < 0) return 0;
    for (int i = 0; i < 9; ++i) R[i] = Rs[9 * best + i];
    for (int i = 0; i < 3; ++i) t[i] = ts[3 * best + i];
    return 1;
}

// ---------------------------------------------------------------- wave64 helpers
MFR_DEV double wave_sum(double v)
{
    // xor butterfly 32,16,8,4,2,1: every lane ends with the same bits (a+b == b+a)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
MFR_DEV int lane_id() { return (int)(threadIdx.x & 63); }

}  // namespace mfr
