// emat_lds.h -- Nister's 5-point solver with its working set in LDS (gfx950).
//
// Same arithmetic, statement for statement, as mfr::fivept (emat_dev.h; restated cv.findEssentialMat minimal solver,
// lib/models/matching/pose_solver.py:46-48) -- the results are bit-identical -- but every array that is indexed at run time
// (Gauss-Jordan pivoting, the monomial index tables, the derivative stack of the root finder) lives in LDS instead of the
// per-lane private segment.  emat_hyp_kernel ran one solve per lane with 5984 bytes of scratch per lane: every array access was
// a trip to the memory hierarchy and a launch of 16 k solves took 9.9 ms whatever the batch (pure latency, one wavefront per
// CU).  Layout: element e of lane l at word e * 64 + l (conflict-free ds_read/write_b64); 290 doubles per lane = 145 KB per
// 64-lane workgroup, by re-using dead regions:
//     [0, 36)    Ep[9][4]        null-space basis (live to the end)
//     [36, 236)  A[5][9] then M[10][20] (built directly in its column-permuted order), then P / c0 / c1 / c2 / roots and the
//                root finder's derivative stack once Bx / By / B1 have been extracted
//     [236, 290) m[10], neg[4], one row of EEt[3][10], tr[10]; later Bx[3][4], By[3][4], B1[3][5]
// The trace-free part of E E^T is formed one row of blocks at a time (the diagonal blocks are evaluated twice, identically),
// which keeps the order of every floating-point operation of the original.
#pragma once
#include "emat_dev.h"

namespace mfr {

#define FP_LDS_DOUBLES 290
#ifndef FP_STAGE_LIMIT
#define FP_STAGE_LIMIT 99           // tools/ubench/fivept_time.hip builds the solver with earlier cut-offs to time its stages
#endif
// a cut-off keeps the work before it observable (checksum of the whole window to global memory), else it is dead code
#define FP_CUT(k) do { if (FP_STAGE_LIMIT < (k)) { double cs = 0.0; for (int q = 0; q < FP_LDS_DOUBLES; ++q) cs += S[q]; Es[0] = cs; return 0; } } while (0)

// compile-time copies of the monomial index tables: with the loops fully unrolled every LDS offset is an immediate (the
// run-time tables of emat_dev.h cost a scalar memory load per multiply-add)
struct FpTables {
    int idx11[4][4] = { {0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9} };
    int idx21[10][4] = { {0, 1, 2, 10}, {1, 3, 4, 11}, {2, 4, 5, 12}, {3, 6, 7, 13}, {4, 7, 8, 14}, {5, 8, 9, 15}, {10, 11, 12, 16},
                         {11, 13, 14, 17}, {12, 14, 15, 18}, {16, 17, 18, 19} };
    // INV[NPERM[c]] = c: M[r][c] = C[r][NPERM[c]]  <=>  C[r][k] = M[r][INV[k]]
    int nperm_inv[20] = { 0, 2, 4, 3, 8, 10, 1, 6, 13, 16, 5, 9, 11, 7, 14, 17, 12, 15, 18, 19 };
};

struct LdsArr {
    double *p;                                        // lane's element 0
    __device__ __forceinline__ double &operator[](int i) const { return p[i * 64]; }
    __device__ __forceinline__ LdsArr at(int off) const { return LdsArr{ p + off * 64 }; }
};

// Horner on a register-resident, zero-padded coefficient vector.  Padding is exact: with c[k] = 0 for k > deg the first steps
// compute (+-0) * x + c[deg] = c[deg] bit for bit (x finite, c[deg] != 0), after which the recurrence is the original one.  The
// coefficients of a derivative level are loaded from LDS ONCE per level instead of once per Horner step: with one wavefront
// per CU a dependent LDS load per step (~100 cycles, 20 steps per Newton iteration, ~60 iterations x ~65 brackets, every lane
// waiting for the slowest) was >90 % of the solve.
template <int N>
MFR_DEV double horner_reg(const double (&c)[N], double x)
{
    double y = c[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) y = y * x + c[i];
    return y;
}

MFR_DEV double reg_refine_root(const double (&c)[11], const double (&dc)[10], double lo, double hi, double flo)
{
    // safeguarded Newton (rtsafe), statement for statement geom_dev.h refine_root
    double x = 0.5 * (lo + hi), dxold = hi - lo, dx = dxold;
    double fx = horner_reg(c, x), dfx = horner_reg(dc, x);
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
        fx = horner_reg(c, x);
        dfx = horner_reg(dc, x);
    }
    return x;
}

// poly_real_roots<10> with d[10][11], crit[12], cur[12] in LDS (ws: 134 doubles); c_in / roots are LDS arrays too.
//
// Control flow: the textbook form (for each derivative level, for each bracket, refine to convergence) makes a wavefront pay,
// for every bracket, the iteration count of its SLOWEST lane -- with 64 independent polynomials per wavefront and up to 200
// safeguarded-Newton steps per bracket that was ~6 ms per launch on LoFTR's correspondences
// (measured on random two-view geometry: 2.8 ms nested -> 2.3 ms as a state machine; a variant with one shared evaluation per
// trip was slower, 3.6 ms).  Here every lane runs the same
// computation as a small state machine (LEVEL: load a level's coefficients, SCAN: evaluate the next bracket end, REFINE: one
// rtsafe step) and the wavefront loops until all lanes are DONE: a lane never waits inside somebody else's bracket, the cost
// is the slowest lane's TOTAL work.  Each lane executes exactly the statement sequence of geom_dev.h poly_real_roots /
// refine_root on its own data, so the roots are bit-identical.
MFR_DEV_NOINLINE int lds_poly_real_roots10(LdsArr c_in, int deg, LdsArr roots, LdsArr ws)
{
    const LdsArr d = ws, crit = ws.at(110), cur = ws.at(122);
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
    for (int i = 0; i <= deg; ++i) d[i] = c_in[i];
    for (int L = 1; L < deg; ++L)
        for (int i = 0; i <= deg - L; ++i) d[L * 11 + i] = d[(L - 1) * 11 + i + 1] * (double)(i + 1);
    int nc = 1;
    crit[0] = -d[(deg - 1) * 11] / d[(deg - 1) * 11 + 1];

    enum { M_LEVEL = 0, M_SCAN = 1, M_REFINE = 2, M_DONE = 3 };
    int mode = (deg >= 2) ? M_LEVEL : M_DONE;
    int L = deg - 2, i = 0, nr = 0, it = 0;
    double pc[11], dpc[10];
    double xl = 0.0, fl = 0.0, xh = 0.0, fh = 0.0;                       // bracket scan state
    double x = 0.0, lo = 0.0, hi = 0.0, dx = 0.0, dxold = 0.0, fx = 0.0, dfx = 0.0, flo = 0.0;   // rtsafe state
    while (__ballot(mode != M_DONE)) {
        if (mode == M_LEVEL) {
            const int m = deg - L;
#pragma unroll
            for (int k = 0; k < 11; ++k) pc[k] = (k <= m) ? d[L * 11 + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 10; ++k) dpc[k] = (k <= m - 1) ? d[(L + 1) * 11 + k] : 0.0;
            nr = 0; i = 0;
            xl = -bound; fl = horner_reg(pc, xl);
            mode = M_SCAN;
        } else if (mode == M_SCAN) {
            if (i > nc) {                                                 // after the bracket loop of this level
                if (fl == 0.0 && (nr == 0 || cur[nr - 1] != xl)) cur[nr++] = xl;
                nc = nr;
                for (int k = 0; k < nr; ++k) crit[k] = cur[k];
                --L;
                mode = (L >= 0) ? M_LEVEL : M_DONE;
            } else {
                xh = (i < nc) ? crit[i] : bound;
                if (i < nc && !(xh > xl)) { ++i; }                        // `continue`
                else {
                    fh = horner_reg(pc, xh);
                    bool advance = true;
                    if (fl == 0.0) {
                        if (nr == 0 || cur[nr - 1] != xl) cur[nr++] = xl;
                    } else if (fh != 0.0 && ((fl < 0.0) != (fh < 0.0))) {
                        // refine_root(p, dp, m, xl, xh, fl): set up, the iterations run in M_REFINE
                        lo = xl; hi = xh; flo = fl;
                        x = 0.5 * (lo + hi); dxold = hi - lo; dx = dxold;
                        fx = horner_reg(pc, x); dfx = horner_reg(dpc, x);
                        it = 0;
                        mode = M_REFINE;
                        advance = false;
                    }
                    if (advance) { xl = xh; fl = fh; ++i; }
                }
            }
        } else if (mode == M_REFINE) {
            bool stop = (it >= 200) || (fx == 0.0);
            if (!stop) {
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
                if (xn == x) stop = true;
                else {
                    const double adx = dx < 0.0 ? -dx : dx, ax = xn < 0.0 ? -xn : xn;
                    x = xn;
                    if (adx <= 2e-16 * ax || adx < 1e-300) stop = true;
                    else { fx = horner_reg(pc, x); dfx = horner_reg(dpc, x); ++it; }
                }
            }
            if (stop) {                                                   // cur[nr++] = refine_root(...); xl = xh; fl = fh
                cur[nr++] = x;
                xl = xh; fl = fh; ++i;
                mode = M_SCAN;
            }
        }
    }
    for (int k = 0; k < nc; ++k) roots[k] = crit[k];
    return nc;
}

MFR_DEV void lds_p_mul11(LdsArr a, LdsArr b, LdsArr o)
{
    constexpr FpTables T{};
    double av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { av[i] = a[i]; bv[i] = b[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[T.idx11[i][j]] = o[T.idx11[i][j]] + av[i] * bv[j];
}
// o is a row of the (column-permuted) M: monomial k lands in column nperm_inv[k]
MFR_DEV void lds_p_mul21_perm(LdsArr a, LdsArr b, LdsArr o)
{
    constexpr FpTables T{};
    double av[10], bv[4];
#pragma unroll
    for (int i = 0; i < 10; ++i) av[i] = a[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = b[j];
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = T.nperm_inv[T.idx21[i][j]]; o[c] = o[c] + av[i] * bv[j]; }
}

// x0, x1: 5 normalised points each (registers); Es: up to 10 E written to GLOBAL memory (row-major, unit Frobenius norm).
// S = this lane's LDS window (FP_LDS_DOUBLES doubles, stride 64), colp = this lane's int window (9 ints, stride 64).
// front_only: stop after the degree-10 polynomial and hand {Ep[36], Bx[12], By[12], B1[15], P[11]} (86 doubles) to the caller in
// Es[0..86) (return -1); the root stage then runs in emat_roots_kernel with 8 lanes per hypothesis.
MFR_DEV int fivept_lds(const double *x0, const double *x1, double *Es, double *lds_lane, int *colp_lane, bool front_only = false)
{
    const LdsArr S{ lds_lane };
    const LdsArr Ep = S, A = S.at(36), M = S.at(36), m = S.at(236), neg = S.at(246), EE = S.at(250), tr = S.at(280);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double a = x0[2 * i], b = x0[2 * i + 1], c = x1[2 * i], d = x1[2 * i + 1];
        const LdsArr Ai = A.at(9 * i);
        Ai[0] = c * a; Ai[1] = c * b; Ai[2] = c; Ai[3] = d * a; Ai[4] = d * b; Ai[5] = d;
        Ai[6] = a; Ai[7] = b; Ai[8] = 1.0;
    }
    for (int j = 0; j < 9; ++j) colp_lane[j * 64] = j;
    for (int r = 0; r < 5; ++r) {
        int pr = r, pc = r;
        double best = -1.0;
        for (int i = r; i < 5; ++i)
            for (int j = r; j < 9; ++j) {
                const double e = A[9 * i + j];
                const double v = e < 0.0 ? -e : e;
                if (v > best) { best = v; pr = i; pc = j; }
            }
        if (!(best > 1e-300)) return 0;
        if (pr != r)
            for (int j = 0; j < 9; ++j) { const double tmp = A[9 * r + j]; A[9 * r + j] = A[9 * pr + j]; A[9 * pr + j] = tmp; }
        if (pc != r) {
            for (int i = 0; i < 5; ++i) { const double tmp = A[9 * i + r]; A[9 * i + r] = A[9 * i + pc]; A[9 * i + pc] = tmp; }
            const int ti = colp_lane[r * 64]; colp_lane[r * 64] = colp_lane[pc * 64]; colp_lane[pc * 64] = ti;
        }
        const double inv = 1.0 / A[9 * r + r];
        double rr[9];                                     // pivot row staged in registers: batched LDS traffic, same arithmetic
#pragma unroll
        for (int j = 0; j < 9; ++j) rr[j] = A[9 * r + j];
#pragma unroll
        for (int j = 0; j < 9; ++j) { rr[j] = rr[j] * inv; A[9 * r + j] = rr[j]; }
        for (int i = 0; i < 5; ++i)
            if (i != r) {
                const double f = A[9 * i + r];
                double ri[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) ri[j] = A[9 * i + j];
#pragma unroll
                for (int j = 0; j < 9; ++j) A[9 * i + j] = ri[j] - f * rr[j];
            }
    }
    for (int k = 0; k < 4; ++k) {
        // v[j] = 0 except v[5 + k] = 1 and v[r] = -A[r][5 + k] (r < 5); Ep[colp[j]][k] = v[j]
        for (int j = 0; j < 9; ++j) {
            double v = 0.0;
            if (j == 5 + k) v = 1.0;
            if (j < 5) v = -A[9 * j + 5 + k];
            Ep[colp_lane[j * 64] * 4 + k] = v;
        }
    }
    FP_CUT(1);
    // ---- constraint matrix, built directly as M (A's region is dead now)
    for (int q = 0; q < 200; ++q) M[q] = 0.0;
#define MFR_MINOR_L(a, b, c, d)                                       \
    do {                                                              \
        for (int q = 0; q < 10; ++q) m[q] = 0.0;                      \
        lds_p_mul11(Ep.at(4 * (a)), Ep.at(4 * (b)), m);               \
        for (int q = 0; q < 4; ++q) neg[q] = -Ep[4 * (c) + q];        \
        lds_p_mul11(neg, Ep.at(4 * (d)), m);                          \
    } while (0)
    MFR_MINOR_L(4, 8, 5, 7); lds_p_mul21_perm(m, Ep.at(0), M);
    MFR_MINOR_L(5, 6, 3, 8); lds_p_mul21_perm(m, Ep.at(4), M);
    MFR_MINOR_L(3, 7, 4, 6); lds_p_mul21_perm(m, Ep.at(8), M);
#undef MFR_MINOR_L
    // trace of E E^T from its three diagonal blocks
    for (int i = 0; i < 3; ++i) {
        const LdsArr Eii = EE.at(10 * i);
        for (int q = 0; q < 10; ++q) Eii[q] = 0.0;
        for (int k = 0; k < 3; ++k) lds_p_mul11(Ep.at(4 * (3 * i + k)), Ep.at(4 * (3 * i + k)), Eii);
    }
    for (int q = 0; q < 10; ++q) tr[q] = (EE[q] + EE[10 + q]) + EE[20 + q];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            const LdsArr Eij = EE.at(10 * j);
            for (int q = 0; q < 10; ++q) Eij[q] = 0.0;
            for (int k = 0; k < 3; ++k) lds_p_mul11(Ep.at(4 * (3 * i + k)), Ep.at(4 * (3 * j + k)), Eij);
        }
        for (int q = 0; q < 10; ++q) EE[10 * i + q] = EE[10 * i + q] - 0.5 * tr[q];
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) lds_p_mul21_perm(EE.at(10 * k), Ep.at(4 * (3 * k + j)), M.at(20 * (1 + 3 * i + j)));
    }
    FP_CUT(2);
    for (int c = 0; c < 10; ++c) {
        int pr = c;
        double best = -1.0;
        for (int i = c; i < 10; ++i) {
            const double e = M[20 * i + c];
            const double v = e < 0.0 ? -e : e;
            if (v > best) { best = v; pr = i; }
        }
        if (!(best > 1e-300)) return 0;
        if (pr != c)
            for (int j = 0; j < 20; ++j) { const double tmp = M[20 * c + j]; M[20 * c + j] = M[20 * pr + j]; M[20 * pr + j] = tmp; }
        const double inv = 1.0 / M[20 * c + c];
        double rc[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) rc[j] = M[20 * c + j];
#pragma unroll
        for (int j = 0; j < 20; ++j) { rc[j] = rc[j] * inv; M[20 * c + j] = rc[j]; }
        for (int i = 0; i < 10; ++i)
            if (i != c) {
                const double f = M[20 * i + c];
                double ri[20];
#pragma unroll
                for (int j = 0; j < 20; ++j) ri[j] = M[20 * i + j];
#pragma unroll
                for (int j = 0; j < 20; ++j) M[20 * i + j] = ri[j] - f * rc[j];
            }
    }
    FP_CUT(3);
    const LdsArr Bx = S.at(236), By = S.at(248), B1 = S.at(260);     // [3][4], [3][4], [3][5] over the dead m / neg / EE / tr region
    for (int r = 0; r < 3; ++r) {
        const LdsArr e = M.at(20 * (4 + 2 * r)), f = M.at(20 * (5 + 2 * r));
        Bx[4 * r] = e[12]; Bx[4 * r + 1] = e[11] - f[12]; Bx[4 * r + 2] = e[10] - f[11]; Bx[4 * r + 3] = -f[10];
        By[4 * r] = e[15]; By[4 * r + 1] = e[14] - f[15]; By[4 * r + 2] = e[13] - f[14]; By[4 * r + 3] = -f[13];
        B1[5 * r] = e[19]; B1[5 * r + 1] = e[18] - f[19]; B1[5 * r + 2] = e[17] - f[18]; B1[5 * r + 3] = e[16] - f[17]; B1[5 * r + 4] = -f[16];
    }
    // M's region is dead: P[11] | c0[8] | c1[8] | c2[7] | roots[10] | root-finder workspace[134]
    const LdsArr P = S.at(36), c0 = S.at(47), c1 = S.at(55), c2 = S.at(63), roots = S.at(70), rws = S.at(80);
    for (int q = 0; q < 11; ++q) P[q] = 0.0;
    for (int q = 0; q < 8; ++q) { c0[q] = 0.0; c1[q] = 0.0; }
    for (int q = 0; q < 7; ++q) c2[q] = 0.0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 5; ++b) {
            c0[a + b] = c0[a + b] + (By[4 + a] * B1[10 + b] - B1[5 + b] * By[8 + a]);
            c1[a + b] = c1[a + b] + (Bx[4 + a] * B1[10 + b] - B1[5 + b] * Bx[8 + a]);
        }
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) c2[a + b] = c2[a + b] + (Bx[4 + a] * By[8 + b] - By[4 + a] * Bx[8 + b]);
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 8; ++b) P[a + b] = P[a + b] + (Bx[a] * c0[b] - By[a] * c1[b]);
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 7; ++b) P[a + b] = P[a + b] + B1[a] * c2[b];
    FP_CUT(4);
    if (front_only) {
        for (int q = 0; q < 36; ++q) Es[q] = Ep[q];
        for (int q = 0; q < 39; ++q) Es[36 + q] = Bx[q];          // Bx | By | B1 are contiguous
        for (int q = 0; q < 11; ++q) Es[75 + q] = P[q];
        return -1;
    }
    const int nr = lds_poly_real_roots10(P, 10, roots, rws);
    if (FP_STAGE_LIMIT < 5) { double cs = (double)nr; for (int q = 0; q < 10; ++q) cs += roots[q]; Es[0] = cs; return 0; }
    int ns = 0;
    for (int r = 0; r < nr; ++r) {
        const double z = roots[r];
        double bx[3], by[3], b1[3];
        for (int k = 0; k < 3; ++k) {
            bx[k] = ((Bx[4 * k + 3] * z + Bx[4 * k + 2]) * z + Bx[4 * k + 1]) * z + Bx[4 * k];
            by[k] = ((By[4 * k + 3] * z + By[4 * k + 2]) * z + By[4 * k + 1]) * z + By[4 * k];
            b1[k] = (((B1[5 * k + 4] * z + B1[5 * k + 3]) * z + B1[5 * k + 2]) * z + B1[5 * k + 1]) * z + B1[5 * k];
        }
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, bestw = -1.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int p = a, q = (a + 1) % 3;
            const double w0 = by[p] * b1[q] - b1[p] * by[q];
            const double w1 = b1[p] * bx[q] - bx[p] * b1[q];
            const double w2 = bx[p] * by[q] - by[p] * bx[q];
            const double aw = w2 < 0.0 ? -w2 : w2;
            if (aw > bestw) { bestw = aw; v0 = w0; v1 = w1; v2 = w2; }
        }
        if (!(bestw > 0.0)) continue;
        const double x = v0 / v2, y = v1 / v2;
        double *E = Es + 9 * ns, nn = 0.0;
        double ev[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            ev[e] = ((x * Ep[4 * e] + y * Ep[4 * e + 1]) + z * Ep[4 * e + 2]) + Ep[4 * e + 3];
            nn = nn + ev[e] * ev[e];
        }
        if (!(nn > 0.0) || !(nn < 1e300)) continue;
        const double s = 1.0 / sqrt(nn);
#pragma unroll
        for (int e = 0; e < 9; ++e) E[e] = ev[e] * s;
        ++ns;
    }
    return ns;
}


// ---------------------------------------------------------------------------------------------------------------------
// Root stage with EIGHT lanes per hypothesis (emat_roots_kernel).  The derivative-isolation root finder is sequential across
// derivative levels but the brackets of one level are independent: the lanes of a group evaluate the bracket ends in parallel
// and refine the sign-change brackets in parallel, so a level costs the iterations of its slowest bracket instead of their
// sum, and 16 k hypotheses become 2000 wavefronts instead of 250 (one latency-bound wavefront per CU before).  Every
// bracket's rtsafe iteration, every Horner evaluation and every derivative coefficient ((c[i+L] * (i+L)) * (i+L-1) ... * (i+1),
// the multiplication order of the derivative tower) is the arithmetic of geom_dev.h poly_real_roots: bit-identical roots.
// ---------------------------------------------------------------------------------------------------------------------
MFR_DEV double fp_dcoef(const double (&c)[11], int L, int i)      // coefficient i of the L-th derivative
{
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) if (k == i + L) v = c[k];
    for (int k = L; k >= 1; --k) v = v * (double)(i + k);
    return v;
}

#define FPR_GROUP 8
#define FPR_HYP_PER_WG 32
struct FprShared {                                  // per hypothesis
    double in[75];                                  // Ep[36] | Bx[12] | By[12] | B1[15]
    double ends[12], fe[12], rr[12], crit[12];
    int ne, nc;
};

MFR_DEV void fp_roots_group(double *slot, int32_t *nsol_out, volatile FprShared *sh, int sub, bool active)
{
    double c[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) c[k] = active ? slot[75 + k] : 0.0;
    if (active) {
        for (int q = sub; q < 75; q += FPR_GROUP) sh->in[q] = slot[q];
    }
    int deg = 10;                                   // while (deg > 0 && c[deg] == 0) --deg, with static register indices
#pragma unroll
    for (int k = 10; k >= 1; --k) if (deg == k && c[k] == 0.0) deg = k - 1;
    double bound = 0.0, cdeg = 0.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) if (k == deg) cdeg = c[k];
    bool live = active && deg > 0;
    if (live) {
        for (int i = 0; i < deg; ++i) {
            double ci = 0.0;
#pragma unroll
            for (int k = 0; k < 11; ++k) if (k == i) ci = c[k];
            double r = ci / cdeg;
            if (r < 0.0) r = -r;
            if (r > bound) bound = r;
        }
        bound = bound + 1.0;
        if (!(bound < 1e300)) live = false;
    }
    int nc = 0;
    if (live) {
        nc = 1;
        if (sub == 0) sh->crit[0] = -fp_dcoef(c, deg - 1, 0) / fp_dcoef(c, deg - 1, 1);
    }
    int L = live ? deg - 2 : -1;
    while (__ballot(L >= 0)) {
        const bool lv = L >= 0;
        double pc[11], dpc[10];
        const int m = deg - L;
#pragma unroll
        for (int k = 0; k < 11; ++k) pc[k] = (lv && k <= m) ? fp_dcoef(c, L, k) : 0.0;
#pragma unroll
        for (int k = 0; k < 10; ++k) dpc[k] = (lv && k <= m - 1) ? fp_dcoef(c, L + 1, k) : 0.0;
        // bracket ends: -bound, the increasing critical points of the level above, +bound (the scan's `continue` rule)
        if (lv && sub == 0) {
            int ne = 0;
            double last = -bound;
            sh->ends[ne++] = last;
            for (int i = 0; i <= nc; ++i) {
                const double xh = (i < nc) ? sh->crit[i] : bound;
                if (i < nc && !(xh > last)) continue;
                sh->ends[ne++] = xh; last = xh;
            }
            sh->ne = ne;
        }
        const int ne = lv ? sh->ne : 0;
        for (int j = sub; j < ne; j += FPR_GROUP) sh->fe[j] = horner_reg(pc, sh->ends[j]);
        for (int j0 = 0; j0 < 12; j0 += FPR_GROUP) {                   // refinements of the sign-change brackets, one per lane
            const int j = j0 + sub;
            bool work = false;
            double lo = 0.0, hi = 0.0, flo = 0.0;
            if (lv && j + 1 < ne) {
                const double fl = sh->fe[j], fh = sh->fe[j + 1];
                if (fl != 0.0 && fh != 0.0 && ((fl < 0.0) != (fh < 0.0))) { work = true; lo = sh->ends[j]; hi = sh->ends[j + 1]; flo = fl; }
            }
            if (!__ballot(work)) continue;
            // reg_refine_root, run in lock step by the lanes that have a bracket (statement for statement the same loop)
            double x = 0.5 * (lo + hi), dxold = hi - lo, dx = dxold;
            double fx = work ? horner_reg(pc, x) : 0.0, dfx = work ? horner_reg(dpc, x) : 1.0;
            int it = 0;
            bool run = work;
            while (__ballot(run)) {
                if (run) {
                    if (it >= 200 || fx == 0.0) run = false;
                    else {
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
                        if (xn == x) run = false;
                        else {
                            const double adx = dx < 0.0 ? -dx : dx, ax = xn < 0.0 ? -xn : xn;
                            x = xn;
                            if (adx <= 2e-16 * ax || adx < 1e-300) run = false;
                            else { fx = horner_reg(pc, x); dfx = horner_reg(dpc, x); ++it; }
                        }
                    }
                }
            }
            if (work) sh->rr[j] = x;
        }
        // the scan's bookkeeping, in bracket order (one lane; the refined roots are ready)
        if (lv && sub == 0) {
            int nr = 0;
            double prev = 0.0;
            for (int j = 0; j + 1 < ne; ++j) {
                const double xl = sh->ends[j], fl = sh->fe[j], fh = sh->fe[j + 1];
                if (fl == 0.0) { if (nr == 0 || prev != xl) { sh->crit[nr++] = xl; prev = xl; } }
                else if (fh != 0.0 && ((fl < 0.0) != (fh < 0.0))) { prev = sh->rr[j]; sh->crit[nr++] = prev; }
            }
            const double xl = sh->ends[ne - 1], fl = sh->fe[ne - 1];
            if (fl == 0.0 && (nr == 0 || prev != xl)) sh->crit[nr++] = xl;
            sh->nc = nr;
        }
        if (lv) { nc = sh->nc; --L; }
    }
    // one E per real root (emat_dev.h fivept tail), roots handled in parallel, compacted in root order
    int ns = 0;
    for (int r0 = 0; r0 < 16; r0 += FPR_GROUP) {
        const int r = r0 + sub;
        bool ok = false;
        double ev[9];
        if (live && r < nc) {
            const double z = sh->crit[r];
            const volatile double *Ep = sh->in, *Bx = sh->in + 36, *By = sh->in + 48, *B1 = sh->in + 60;
            double bx[3], by[3], b1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                bx[k] = ((Bx[4 * k + 3] * z + Bx[4 * k + 2]) * z + Bx[4 * k + 1]) * z + Bx[4 * k];
                by[k] = ((By[4 * k + 3] * z + By[4 * k + 2]) * z + By[4 * k + 1]) * z + By[4 * k];
                b1[k] = (((B1[5 * k + 4] * z + B1[5 * k + 3]) * z + B1[5 * k + 2]) * z + B1[5 * k + 1]) * z + B1[5 * k];
            }
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, bestw = -1.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int p = a, q = (a + 1) % 3;
                const double w0 = by[p] * b1[q] - b1[p] * by[q];
                const double w1 = b1[p] * bx[q] - bx[p] * b1[q];
                const double w2 = bx[p] * by[q] - by[p] * bx[q];
                const double aw = w2 < 0.0 ? -w2 : w2;
                if (aw > bestw) { bestw = aw; v0 = w0; v1 = w1; v2 = w2; }
            }
            if (bestw > 0.0) {
                const double x = v0 / v2, y = v1 / v2;
                double nn = 0.0;
#pragma unroll
                for (int e = 0; e < 9; ++e) {
                    ev[e] = ((x * Ep[4 * e] + y * Ep[4 * e + 1]) + z * Ep[4 * e + 2]) + Ep[4 * e + 3];
                    nn = nn + ev[e] * ev[e];
                }
                if ((nn > 0.0) && (nn < 1e300)) {
                    const double s = 1.0 / sqrt(nn);
#pragma unroll
                    for (int e = 0; e < 9; ++e) ev[e] = ev[e] * s;
                    ok = true;
                }
            }
        }
        const unsigned long long bal = __ballot(ok);
        const int lane = (int)(threadIdx.x & 63), gbase = lane & ~(FPR_GROUP - 1);
        const unsigned seg = (unsigned)((bal >> gbase) & 0xffull);
        if (ok) {
            double *E = slot + 9 * (ns + __popc(seg & ((1u << sub) - 1u)));
#pragma unroll
            for (int e = 0; e < 9; ++e) E[e] = ev[e];
        }
        ns += __popc(seg);
    }
    if (active && sub == 0) *nsol_out = ns;
}

}  // namespace mfr
