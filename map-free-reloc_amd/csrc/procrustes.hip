// procrustes.hip -- 3-D/3-D Procrustes path of the relative-pose solver on gfx950 (MI355X).
//
// Replaces ProcrustesSolver.estimate_pose (lib/models/matching/pose_solver.py:238-320, REFINE False as
// in config/matching/mapfree/sg_procrustes_dptkitti.yaml) for a batch of pairs:
//   proc_lift_kernel    np.int32 both views (:248-249), depth gather (:256-258), valid vs each map's
//                       minimum (:261, Q6), back-project both (:273-274), ordered compaction
//   proc_hyp_kernel     o3d registration_ransac_based_on_correspondence (:286-287): lane per 3-point
//                       Kabsch (Horn quaternion + fixed-sweep Jacobi), wavefront per hypothesis
//                       scoring (count + squared-error sum in wave64 order)
//   proc_select_kernel  replay of Open3D's best-model rule (fitness, then inlier RMSE) and its
//                       confidence-based exit, final re-fit on the inliers, inliers = int(fitness*N)
// Open3D 0.17 is not available offline: restated from the published algorithm, parity unpinned vs
// Open3D (see oracle/mfr_oracle_procrustes.c for the substitutions).  -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "geom_dev.h"

using namespace mfr;
#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)
#define PR_NSEG 16
#define PR_TILE 512

MFR_DEV void jacobi4_maxvec(double A[4][4], double q[4])
{
    double V[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 10; ++sweep)
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                const double apq = A[p][r];
                if (apq == 0.0) continue;
                const double theta = (A[r][r] - A[p][p]) / (2.0 * apq);
                const double at = theta < 0.0 ? -theta : theta;
                double t = 1.0 / (at + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][r]; A[k][p] = c * akp - s * akq; A[k][r] = s * akp + c * akq; }
                for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[r][k]; A[p][k] = c * apk - s * aqk; A[r][k] = s * apk + c * aqk; }
                for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][r]; V[k][p] = c * vkp - s * vkq; V[k][r] = s * vkp + c * vkq; }
            }
    int b = 0;
    for (int i = 1; i < 4; ++i) if (A[i][i] > A[b][b]) b = i;
    const double nn = sqrt(((V[0][b] * V[0][b] + V[1][b] * V[1][b]) + V[2][b] * V[2][b]) + V[3][b] * V[3][b]);
    for (int i = 0; i < 4; ++i) q[i] = V[i][b] / nn;
}

MFR_DEV_NOINLINE void kabsch_from_moments(const double *s, double *R, double *t)
{
    const double n = s[0], pc[3] = { s[1] / n, s[2] / n, s[3] / n }, qc[3] = { s[4] / n, s[5] / n, s[6] / n };
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
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i) t[i] = qc[i] - ((R[3 * i] * pc[0] + R[3 * i + 1] * pc[1]) + R[3 * i + 2] * pc[2]);
}

MFR_DEV double dist2(const double *R, const double *t, double p0, double p1, double p2, double q0, double q1, double q2)
{
    const double d0 = (((R[0] * p0 + R[1] * p1) + R[2] * p2) + t[0]) - q0;
    const double d1 = (((R[3] * p0 + R[4] * p1) + R[5] * p2) + t[1]) - q1;
    const double d2 = (((R[6] * p0 + R[7] * p1) + R[8] * p2) + t[2]) - q2;
    return (d0 * d0 + d1 * d1) + d2 * d2;
}

MFR_DEV void sample3_model(const double *P, const double *Q, int n, uint64_t seed, uint64_t pair_id, int it, double *R, double *t)
{
    int s[3];
    if (n == 3) { s[0] = 0; s[1] = 1; s[2] = 2; } else sample_distinct<3>(seed, pair_id, (uint32_t)it, n, s);
    double m[16];
    for (int k = 0; k < 16; ++k) m[k] = 0.0;
    for (int j = 0; j < 3; ++j) {
        const double *p = P + 3 * (size_t)s[j], *q = Q + 3 * (size_t)s[j];
        m[0] = m[0] + 1.0;
        for (int a = 0; a < 3; ++a) { m[1 + a] = m[1 + a] + p[a]; m[4 + a] = m[4 + a] + q[a]; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) m[7 + 3 * a + b] = m[7 + 3 * a + b] + p[a] * q[b];
    }
    kabsch_from_moments(m, R, t);
}

__global__ void __launch_bounds__(256) proc_lift_kernel(
    const float *__restrict__ pts0, const float *__restrict__ pts1, const int32_t *__restrict__ n_corr, int maxN,
    const float *__restrict__ depth0, const float *__restrict__ depth1, const float *__restrict__ pmin0,
    const float *__restrict__ pmin1, int H, int W, const void *__restrict__ K0, const void *__restrict__ K1, int k_dtype,
    double *__restrict__ P, double *__restrict__ Q, int32_t *__restrict__ n_valid)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    float m0 = pmin0[b * PR_NSEG], m1 = pmin1[b * PR_NSEG];
    for (int s = 1; s < PR_NSEG; ++s) {
        const float a = pmin0[b * PR_NSEG + s], c = pmin1[b * PR_NSEG + s];
        if (a < m0) m0 = a;
        if (c < m1) m1 = c;
    }
    double Ki0[4], Ki1[4];
    kinv(K0, k_dtype, b, Ki0); kinv(K1, k_dtype, b, Ki1);
    const float *p0 = pts0 + (size_t)b * maxN * 2, *p1 = pts1 + (size_t)b * maxN * 2;
    const float *d0m = depth0 + (size_t)b * H * W, *d1m = depth1 + (size_t)b * H * W;
    double *oP = P + (size_t)b * maxN * 3, *oQ = Q + (size_t)b * maxN * 3;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 256) {
        const int i = start + tid;
        bool valid = false;
        int u0 = 0, v0 = 0, u1 = 0, v1 = 0;
        float d0 = 0.f, d1 = 0.f;
        if (i < n) {
            u0 = (int)p0[2 * i]; v0 = (int)p0[2 * i + 1]; u1 = (int)p1[2 * i]; v1 = (int)p1[2 * i + 1];
            if (u0 >= 0 && u0 < W && v0 >= 0 && v0 < H && u1 >= 0 && u1 < W && v1 >= 0 && v1 < H) {
                d0 = d0m[v0 * W + u0]; d1 = d1m[v1 * W + u1];
                valid = (d0 > m0) && (d1 > m1);
            }
        }
        const unsigned long long bal = __ballot(valid);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wid; ++w) off += wave_cnt[w];
        if (valid) {
            const int m = off + wpre;
            double a[3], c[3];
            backproject(u0, v0, d0, Ki0, a); backproject(u1, v1, d1, Ki1, c);
            oP[3 * m] = a[0]; oP[3 * m + 1] = a[1]; oP[3 * m + 2] = a[2];
            oQ[3 * m] = c[0]; oQ[3 * m + 1] = c[1]; oQ[3 * m + 2] = c[2];
        }
        __syncthreads();
        if (tid == 0) base_s = off + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_valid[b] = base_s;
}

// grid (ceil(iters/256), B)
__global__ void __launch_bounds__(256) proc_hyp_kernel(
    const double *__restrict__ P, const double *__restrict__ Q, const int32_t *__restrict__ n_valid, int maxN,
    int max_iters, double thr2, uint64_t seed, const int64_t *__restrict__ pair_ids,
    int32_t *__restrict__ counts, double *__restrict__ err2)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *model = (double *)smem_raw;                      // [12][256]
    double *tp = model + 12 * 256;                           // SoA tile: 6 x PR_TILE
    int *cnt = (int *)(tp + 6 * PR_TILE);                    // [256]
    double *esum = (double *)(cnt + 256);                    // [256]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = n_valid[b];
    const int it = blockIdx.x * 256 + tid;
    if (n < 3) {
        if (it < max_iters) { counts[(size_t)b * max_iters + it] = -1; err2[(size_t)b * max_iters + it] = 0.0; }
        return;
    }
    const double *Pb = P + (size_t)b * maxN * 3, *Qb = Q + (size_t)b * maxN * 3;
    {
        double R[9], t[3];
        if (it < max_iters) sample3_model(Pb, Qb, n, seed, (uint64_t)pair_ids[b], it, R, t);
        else { for (int k = 0; k < 9; ++k) R[k] = 0.0; for (int k = 0; k < 3; ++k) t[k] = 0.0; }
        for (int k = 0; k < 9; ++k) model[k * 256 + tid] = R[k];
        for (int k = 0; k < 3; ++k) model[(9 + k) * 256 + tid] = t[k];
        cnt[tid] = 0; esum[tid] = 0.0;
    }
    // error sums: wave64 order inside each 512-point tile, tile sums added in tile order (the oracle's order)
    for (int base = 0; base < n; base += PR_TILE) {
        const int tn = min(PR_TILE, n - base);
        __syncthreads();
        for (int i = tid; i < tn; i += 256) {
            const double *p = Pb + 3 * (size_t)(base + i), *q = Qb + 3 * (size_t)(base + i);
            tp[i] = p[0]; tp[PR_TILE + i] = p[1]; tp[2 * PR_TILE + i] = p[2];
            tp[3 * PR_TILE + i] = q[0]; tp[4 * PR_TILE + i] = q[1]; tp[5 * PR_TILE + i] = q[2];
        }
        __syncthreads();
        for (int h = 0; h < 64; ++h) {
            const int hi = wid * 64 + h;
            if (blockIdx.x * 256 + hi >= max_iters) break;
            double R[9], t[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = model[k * 256 + hi];
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = model[(9 + k) * 256 + hi];
            int c = 0;
            double e = 0.0;                                   // this lane's partial over points i with (i & 63) == lane
            for (int i0 = 0; i0 < tn; i0 += 64) {
                const int i = i0 + lane;
                bool in = false;
                double d = 0.0;
                if (i < tn) {
                    d = dist2(R, t, tp[i], tp[PR_TILE + i], tp[2 * PR_TILE + i], tp[3 * PR_TILE + i], tp[4 * PR_TILE + i], tp[5 * PR_TILE + i]);
                    in = d < thr2;
                }
                if (in) e = e + d;
                c += __popcll(__ballot(in));
            }
            const double ts = wave_sum(e);
            if (lane == 0) { cnt[hi] += c; esum[hi] = esum[hi] + ts; }
        }
    }
    __syncthreads();
    if (it < max_iters) { counts[(size_t)b * max_iters + it] = cnt[tid]; err2[(size_t)b * max_iters + it] = esum[tid]; }
}

// one wavefront per pair
__global__ void __launch_bounds__(64) proc_select_kernel(
    const double *__restrict__ P, const double *__restrict__ Q, const int32_t *__restrict__ n_valid,
    const int32_t *__restrict__ n_corr, int maxN, int max_iters, double thr2, double conf, uint64_t seed,
    const int64_t *__restrict__ pair_ids, const int32_t *__restrict__ counts, const double *__restrict__ err2,
    int32_t *__restrict__ idx_ws, double *__restrict__ Rout, double *__restrict__ tout, int32_t *__restrict__ n_inliers,
    int32_t *__restrict__ status, int32_t *__restrict__ best_iter, int32_t *__restrict__ iters_run)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = n_valid[b];
    const double *Pb = P + (size_t)b * maxN * 3, *Qb = Q + (size_t)b * maxN * 3;
    int32_t *idx = idx_ws + (size_t)b * maxN;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    int st = MFR_ST_OK;
    if (n_corr[b] < 3) st = MFR_ST_TOO_FEW;                                 // :252-253
    else if (n < 3) st = MFR_ST_BAD_DEPTH;                                  // :262-263
    double R[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
    int bit = -1, bcnt = 0, run = 0, cntf = 0;
    if (st == MFR_ST_OK) {
        // sequential replay of Open3D's loop (every lane runs it redundantly on broadcast loads)
        const int32_t *cn = counts + (size_t)b * max_iters;
        const double *er = err2 + (size_t)b * max_iters;
        double berr = 0.0;
        int est_k = max_iters, it = 0;
        const double lnum = det_log(1.0 - conf);
        for (it = 0; it < est_k; ++it) {
            const int c = cn[it];
            const double e = er[it];
            bool better = false;
            if (c > bcnt) better = true;
            else if (c == bcnt && c > 0 && e < berr) better = true;
            if (better) {
                bcnt = c; berr = e; bit = it;
                const double ratio = (double)c / (double)n, r3 = (ratio * ratio) * ratio;
                int k;
                if (r3 >= 1.0) k = 0;
                else {
                    const double kd = lnum / det_log(1.0 - r3);
                    k = (kd < (double)est_k) ? (int)__builtin_ceil(kd) : est_k;
                }
                if (k < est_k) est_k = k;
            }
        }
        run = it;
        if (bit >= 0) {
            sample3_model(Pb, Qb, n, seed, (uint64_t)pair_ids[b], bit, R, t);
            int m = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const bool in = (i < n) && (dist2(R, t, Pb[3 * (size_t)i], Pb[3 * (size_t)i + 1], Pb[3 * (size_t)i + 2],
                                                  Qb[3 * (size_t)i], Qb[3 * (size_t)i + 1], Qb[3 * (size_t)i + 2]) < thr2);
                const unsigned long long bal = __ballot(in);
                if (in) idx[m + __popcll(bal & ((1ull << lane) - 1ull))] = i;
                m += __popcll(bal);
            }
            __threadfence();
            double acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.0;
            for (int q = lane; q < m; q += 64) {
                const int i = idx[q];
                const double *p = Pb + 3 * (size_t)i, *qq = Qb + 3 * (size_t)i;
                acc[0] = acc[0] + 1.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) { acc[1 + c] = acc[1 + c] + p[c]; acc[4 + c] = acc[4 + c] + qq[c]; }
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[7 + 3 * c + d] = acc[7 + 3 * c + d] + p[c] * qq[d];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = wave_sum(acc[k]);
            if (m >= 3) kabsch_from_moments(acc, R, t);
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const bool in = (i < n) && (dist2(R, t, Pb[3 * (size_t)i], Pb[3 * (size_t)i + 1], Pb[3 * (size_t)i + 2],
                                                  Qb[3 * (size_t)i], Qb[3 * (size_t)i + 1], Qb[3 * (size_t)i + 2]) < thr2);
                cntf += __popcll(__ballot(in));
            }
        }
    }
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) Rout[9 * b + k] = (st == MFR_ST_OK) ? R[k] : qnan;
        for (int k = 0; k < 3; ++k) tout[3 * b + k] = (st == MFR_ST_OK) ? t[k] : qnan;
        n_inliers[b] = (st == MFR_ST_OK) ? cntf : 0;
        status[b] = st;
        if (best_iter) best_iter[b] = bit;
        if (iters_run) iters_run[b] = run;
    }
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
struct PrWs { size_t pm0, pm1, P, Q, nvalid, counts, err2, idx, total; };
static PrWs pr_ws_layout(int B, int maxN, int iters)
{
    PrWs w; size_t o = 0;
    w.pm0 = o;    o = align_up(o + sizeof(float) * PR_NSEG * (size_t)B, 256);
    w.pm1 = o;    o = align_up(o + sizeof(float) * PR_NSEG * (size_t)B, 256);
    w.P = o;      o = align_up(o + sizeof(double) * 3 * (size_t)B * maxN, 256);
    w.Q = o;      o = align_up(o + sizeof(double) * 3 * (size_t)B * maxN, 256);
    w.nvalid = o; o = align_up(o + sizeof(int32_t) * (size_t)B, 256);
    w.counts = o; o = align_up(o + sizeof(int32_t) * (size_t)B * iters, 256);
    w.err2 = o;   o = align_up(o + sizeof(double) * (size_t)B * iters, 256);
    w.idx = o;    o = align_up(o + sizeof(int32_t) * (size_t)B * maxN, 256);
    w.total = o;
    return w;
}

extern "C" {

size_t mfr_procrustes_workspace_bytes(int B, int maxN, int max_iters)
{
    if (B <= 0 || maxN <= 0) return 0;
    if (max_iters < 1) max_iters = 1;
    return pr_ws_layout(B, maxN, max_iters).total;
}

int mfr_procrustes_solve_batch(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                               const float *depth0, const float *depth1, int H, int W, const void *K0, const void *K1, int k_dtype,
                               double max_corr_dist, double confidence, int max_iters, uint64_t seed, const int64_t *pair_ids,
                               void *workspace, size_t workspace_bytes, double *R, double *t, int32_t *n_inliers,
                               int32_t *status, int32_t *best_iter, int32_t *iters_run, int32_t *counts_out, void *stream)
{
    if (!pts0 || !pts1 || !n_corr || !depth0 || !depth1 || !K0 || !K1 || !pair_ids || !workspace || !R || !t || !n_inliers ||
        !status || B <= 0 || maxN <= 0 || H <= 0 || W <= 0 || !(max_corr_dist > 0.0) || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    if (max_iters < 1) max_iters = 1;
    const PrWs w = pr_ws_layout(B, maxN, max_iters);
    if (workspace_bytes < w.total) return MFR_E_WORKSPACE;
    char *ws = (char *)workspace;
    hipStream_t s = (hipStream_t)stream;
    float *pm0 = (float *)(ws + w.pm0), *pm1 = (float *)(ws + w.pm1);
    double *P = (double *)(ws + w.P), *Q = (double *)(ws + w.Q), *err2 = (double *)(ws + w.err2);
    int32_t *nvalid = (int32_t *)(ws + w.nvalid), *counts = (int32_t *)(ws + w.counts), *idx = (int32_t *)(ws + w.idx);
    int rc = mfr_depth_min(depth0, B, H, W, pm0, stream);
    if (rc) return rc;
    rc = mfr_depth_min(depth1, B, H, W, pm1, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(proc_lift_kernel, dim3(B), dim3(256), 0, s, pts0, pts1, n_corr, maxN, depth0, depth1, pm0, pm1, H, W, K0,
                       K1, k_dtype, P, Q, nvalid);
    CHECK_LAUNCH();
    const double thr2 = max_corr_dist * max_corr_dist;
    const size_t smem = (size_t)(12 * 256 + 6 * PR_TILE) * sizeof(double) + 256 * sizeof(int) + 256 * sizeof(double);
    hipLaunchKernelGGL(proc_hyp_kernel, dim3((max_iters + 255) / 256, B), dim3(256), smem, s, P, Q, nvalid, maxN, max_iters, thr2,
                       seed, pair_ids, counts, err2);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(proc_select_kernel, dim3(B), dim3(64), 0, s, P, Q, nvalid, n_corr, maxN, max_iters, thr2, confidence, seed,
                       pair_ids, counts, err2, idx, R, t, n_inliers, status, best_iter, iters_run);
    CHECK_LAUNCH();
    if (counts_out)
        if (hipMemcpyAsync(counts_out, counts, sizeof(int32_t) * (size_t)B * max_iters, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return MFR_E_LAUNCH;
    return 0;
}

}  // extern "C"
