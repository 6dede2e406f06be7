// procrustes_icp.hip -- whole-cloud point-to-point ICP refinement of the Procrustes solver on gfx950 (MI355X).
//
// Replaces the PROCRUSTES.REFINE branch of ProcrustesSolver.estimate_pose (lib/models/matching/pose_solver.py:290-319;
// config/matching/scannet/*_icp.yaml): both depth maps back-projected completely (valid = depth > 0, :296-300),
// o3d.pipelines.registration.registration_icp(pcl_0, pcl_1, MAX_CORR_DIST, init, ICPConvergenceCriteria(1e-4, 1e-4, 30))
// (:307-315), inliers = int(fitness * |pcl_1|) (:319).  Open3D is not available offline: its published algorithm is
// restated (oracle/mfr_oracle_icp.c lists the substitutions; parity unpinned vs Open3D, bit-exact vs the oracle).
//
// Mapping: the target cloud is ORGANISED (one point per pixel of depth1), so the KD-tree query "nearest target point within
// r of Y" becomes an exact search over the pixel window that can hold such a point.  For a target point T = (x, y, z) with
// |T - Y| < r:  u_T - u_Y = fx ((x - X)/z + X (Z - z)/(z Z)),  |x - X| < r, |Z - z| < r, z > Z - r  =>
//     |u_T - u_Y| < r (fx + |u_Y - cx|) / (Z - r)            (same for v),
// a few tens of pixels at indoor depths; candidates are pre-filtered on the f32 depth map alone (|z - Z| < r: 4 bytes per
// candidate, L2-resident) before the 24-byte point is touched.  Points with Z <= 2r scan the whole map.
//   icp_prep_kernel    target cloud (f64, [H*W,3]) + |source|, |target| counts
//   icp_assoc_kernel   one thread per source pixel: transform, window search, 17 moments {1, Y, q, Y q^T, d^2};
//                      wave64 butterflies, 4 wave sums added in order -> one partial per 256-pixel block
//   icp_update_kernel  one wavefront per pair: block partials summed in block order, fitness / rmse, Open3D's convergence
//                      test, Horn/Kabsch update composed onto the transform; a `done` flag turns later passes into no-ops
// so the <= 31 evaluate / update rounds are issued without any host round trip.  -ffp-contract=off (bit-exact contract).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"
#include "geom_dev.h"

using namespace mfr;
#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)
#define ICP_NACC 17
#define ICP_STATE 24          // doubles per pair: R[9] t[3] fit_prev rmse_prev done iters fit rmse nS nT ...

// identical arithmetic to procrustes.hip / oracle kabsch_from_moments (Horn quaternion, fixed-sweep Jacobi)
MFR_DEV void icp_jacobi4_maxvec(double A[4][4], double q[4])
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

MFR_DEV_NOINLINE void icp_kabsch_from_moments(const double *s, double *R, double *t)
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
    icp_jacobi4_maxvec(N, q);
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z);       R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);       R[7] = 2.0 * (y * z + w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i) t[i] = qc[i] - ((R[3 * i] * pc[0] + R[3 * i + 1] * pc[1]) + R[3 * i + 2] * pc[2]);
}

// grid (ceil(HW/256), B)
__global__ void __launch_bounds__(256) icp_prep_kernel(const float *__restrict__ depth0, const float *__restrict__ depth1, int HW, int W,
                                                       const void *__restrict__ K1, int k_dtype, const double *__restrict__ Rin, const double *__restrict__ tin,
                                                       const int32_t *__restrict__ status, double *__restrict__ Tc, int32_t *__restrict__ cnt,
                                                       double *__restrict__ state)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < ICP_STATE) {
        double v = 0.0;
        const int k = threadIdx.x;
        if (k < 9) v = Rin[9 * b + k];
        else if (k < 12) v = tin[3 * b + (k - 9)];
        else if (k == 14) v = (status && status[b] != MFR_ST_OK) ? 1.0 : 0.0;       // failed pairs: every pass is a no-op
        state[(size_t)b * ICP_STATE + k] = v;
    }
    bool s = false, t = false;
    if (i < HW) {
        s = depth0[(size_t)b * HW + i] > 0.f;
        const float d = depth1[(size_t)b * HW + i];
        t = d > 0.f;
        double q[3] = { 0.0, 0.0, 0.0 };
        if (t) {
            double Ki[4];
            kinv(K1, k_dtype, b, Ki);
            backproject(i % W, i / W, d, Ki, q);
        }
        double *o = Tc + ((size_t)b * HW + i) * 3;
        o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
    }
    const int ns = __popcll(__ballot(s)), nt = __popcll(__ballot(t));
    if ((threadIdx.x & 63) == 0) {                                                   // integer counts: order-independent
        if (ns) atomicAdd(cnt + 2 * b, ns);
        if (nt) atomicAdd(cnt + 2 * b + 1, nt);
    }
}

// grid (ceil(HW/256), B)
__global__ void __launch_bounds__(256) icp_assoc_kernel(const float *__restrict__ depth0, const float *__restrict__ depth1,
                                                        const double *__restrict__ Tc, int H, int W, const void *__restrict__ K0,
                                                        const void *__restrict__ K1, int k_dtype, double r, const double *__restrict__ state,
                                                        double *__restrict__ partial)
{
    __shared__ double ws[4][ICP_NACC];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const double *st = state + (size_t)b * ICP_STATE;
    if (st[14] != 0.0) return;                                                       // converged / failed pair
    const int HW = H * W, i = blockIdx.x * 256 + tid;
    const float *d0m = depth0 + (size_t)b * HW, *d1m = depth1 + (size_t)b * HW;
    const double *Tb = Tc + (size_t)b * HW * 3;
    double c[ICP_NACC];
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) c[k] = 0.0;
    float d = 0.f;
    if (i < HW) d = d0m[i];
    if (d > 0.f) {
        double R[9], t[3], X[3], Y[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = st[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = st[9 + k];
        double Ki[4];
        kinv(K0, k_dtype, b, Ki);
        backproject(i % W, i / W, d, Ki, X);
        rot_apply(R, t, X, Y);
        double Kd1[4];
        kparams(K1, k_dtype, b, Kd1);
        const double fx = Kd1[0], fy = Kd1[1], cx = Kd1[2], cy = Kd1[3];
        const double Z = Y[2];
        int u0 = 0, u1 = W - 1, v0 = 0, v1 = H - 1;
        bool search = true;
        if (Z > 2.0 * r) {
            const double uS = fx * (Y[0] / Z) + cx, vS = fy * (Y[1] / Z) + cy;
            double au = uS - cx, av = vS - cy;
            if (au < 0.0) au = -au;
            if (av < 0.0) av = -av;
            const double wu = __builtin_ceil((r * (fx + au)) / (Z - r)) + 1.0, wv = __builtin_ceil((r * (fy + av)) / (Z - r)) + 1.0;
            double lo = __builtin_floor(uS) - wu, hi = __builtin_ceil(uS) + wu;
            if (lo < 0.0) lo = 0.0;
            if (hi > (double)(W - 1)) hi = (double)(W - 1);
            if (!(lo <= hi)) search = false;
            else { u0 = (int)lo; u1 = (int)hi; }
            lo = __builtin_floor(vS) - wv; hi = __builtin_ceil(vS) + wv;
            if (lo < 0.0) lo = 0.0;
            if (hi > (double)(H - 1)) hi = (double)(H - 1);
            if (!(lo <= hi)) search = false;
            else { v0 = (int)lo; v1 = (int)hi; }
        } else if (!(Z == Z)) search = false;
        if (search) {
            int best = -1;
            double bd = r * r;
            for (int v = v0; v <= v1; ++v)
                for (int u = u0; u <= u1; ++u) {
                    const float zt = d1m[v * W + u];
                    if (!(zt > 0.f)) continue;
                    double dz = (double)zt - Z;
                    if (dz < 0.0) dz = -dz;
                    if (!(dz < r)) continue;
                    const double *q = Tb + 3 * (size_t)(v * W + u);
                    const double e0 = Y[0] - q[0], e1 = Y[1] - q[1], e2 = Y[2] - q[2];
                    const double dd = (e0 * e0 + e1 * e1) + e2 * e2;
                    if (dd < bd) { bd = dd; best = v * W + u; }
                }
            if (best >= 0) {
                const double *q = Tb + 3 * (size_t)best;
                c[0] = 1.0;
#pragma unroll
                for (int e = 0; e < 3; ++e) { c[1 + e] = Y[e]; c[4 + e] = q[e]; }
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int f = 0; f < 3; ++f) c[7 + 3 * e + f] = Y[e] * q[f];
                c[16] = bd;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) {
        const double s = wave_sum(c[k]);
        if (lane == 0) ws[wid][k] = s;
    }
    __syncthreads();
    if (tid < ICP_NACC)
        partial[((size_t)b * gridDim.x + blockIdx.x) * ICP_NACC + tid] = ((ws[0][tid] + ws[1][tid]) + ws[2][tid]) + ws[3][tid];
}

// grid (B), one wavefront
__global__ void __launch_bounds__(64) icp_update_kernel(const double *__restrict__ partial, int nblk, const int32_t *__restrict__ cnt, int k,
                                                        int max_iter, double rel_fitness, double rel_rmse, double *__restrict__ state)
{
    __shared__ double tot[ICP_NACC];
    const int b = blockIdx.x, lane = threadIdx.x;
    double *st = state + (size_t)b * ICP_STATE;
    if (st[14] != 0.0) return;
    if (lane < ICP_NACC) {
        double a = 0.0;
        const double *p = partial + (size_t)b * nblk * ICP_NACC + lane;
        for (int j = 0; j < nblk; ++j) a = a + p[(size_t)j * ICP_NACC];
        tot[lane] = a;
    }
    __syncthreads();
    if (lane != 0) return;
    const int nS = cnt[2 * b];
    const double n = tot[0];
    const double fit = (nS > 0) ? n / (double)nS : 0.0;
    const double rmse = (n > 0.0) ? sqrt(tot[16] / n) : 0.0;
    st[16] = fit; st[17] = rmse; st[15] = (double)k;
    if (k >= 1) {
        double df = st[12] - fit, dr = st[13] - rmse;
        if (df < 0.0) df = -df;
        if (dr < 0.0) dr = -dr;
        if (df < rel_fitness && dr < rel_rmse) { st[14] = 1.0; return; }
    }
    if (k >= max_iter) { st[14] = 1.0; return; }
    if (n >= 3.0) {
        double m[16], U[9], Ut[3], R[9], t[3], Rn[9], tn[3];
        for (int j = 0; j < 16; ++j) m[j] = tot[j];
        for (int j = 0; j < 9; ++j) R[j] = st[j];
        for (int j = 0; j < 3; ++j) t[j] = st[9 + j];
        icp_kabsch_from_moments(m, U, Ut);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Rn[3 * i + j] = (U[3 * i] * R[j] + U[3 * i + 1] * R[3 + j]) + U[3 * i + 2] * R[6 + j];
            tn[i] = ((U[3 * i] * t[0] + U[3 * i + 1] * t[1]) + U[3 * i + 2] * t[2]) + Ut[i];
        }
        for (int j = 0; j < 9; ++j) st[j] = Rn[j];
        for (int j = 0; j < 3; ++j) st[9 + j] = tn[j];
    }
    st[12] = fit; st[13] = rmse;
}

__global__ void __launch_bounds__(64) icp_finish_kernel(const double *__restrict__ state, const int32_t *__restrict__ cnt, const int32_t *__restrict__ status,
                                                        int B, double *__restrict__ R, double *__restrict__ t, int32_t *__restrict__ n_inliers,
                                                        double *__restrict__ fitness, double *__restrict__ rmse, int32_t *__restrict__ iters)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    if (status && status[b] != MFR_ST_OK) {                                         // untouched pose (NaN from the RANSAC stage), 0 inliers
        n_inliers[b] = 0;
        if (fitness) fitness[b] = 0.0;
        if (rmse) rmse[b] = 0.0;
        if (iters) iters[b] = 0;
        return;
    }
    const double *st = state + (size_t)b * ICP_STATE;
    for (int k = 0; k < 9; ++k) R[9 * b + k] = st[k];
    for (int k = 0; k < 3; ++k) t[3 * b + k] = st[9 + k];
    n_inliers[b] = (int)(st[16] * (double)cnt[2 * b + 1]);                          // :319 int(res.fitness * len(pcl_1.points))
    if (fitness) fitness[b] = st[16];
    if (rmse) rmse[b] = st[17];
    if (iters) iters[b] = (int)st[15];
}

static inline size_t icp_align(size_t x) { return (x + 255) / 256 * 256; }

extern "C" {

size_t mfr_procrustes_icp_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t hw = (size_t)H * W, nblk = (hw + 255) / 256;
    return icp_align(sizeof(double) * 3 * hw * B) + icp_align(sizeof(double) * ICP_NACC * nblk * B) + icp_align(sizeof(double) * ICP_STATE * B) +
           icp_align(sizeof(int32_t) * 2 * B);
}

int mfr_procrustes_icp_refine(const float *depth0, const float *depth1, int B, int H, int W, const void *K0, const void *K1, int k_dtype,
                              double max_corr_dist, double rel_fitness, double rel_rmse, int max_iter, const int32_t *status,
                              void *workspace, size_t workspace_bytes, double *R, double *t, int32_t *n_inliers, double *fitness,
                              double *rmse, int32_t *iters, void *stream)
{
    if (!depth0 || !depth1 || !K0 || !K1 || !workspace || !R || !t || !n_inliers || B <= 0 || H <= 0 || W <= 0 || !(max_corr_dist > 0.0) ||
        max_iter < 0 || (size_t)H * W > 0x3fffffffu || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    if (workspace_bytes < mfr_procrustes_icp_workspace_bytes(B, H, W)) return MFR_E_WORKSPACE;
    const int HW = H * W, nblk = (HW + 255) / 256;
    char *ws = (char *)workspace;
    double *Tc = (double *)ws;                 ws += icp_align(sizeof(double) * 3 * (size_t)HW * B);
    double *partial = (double *)ws;            ws += icp_align(sizeof(double) * ICP_NACC * (size_t)nblk * B);
    double *state = (double *)ws;              ws += icp_align(sizeof(double) * ICP_STATE * B);
    int32_t *cnt = (int32_t *)ws;
    hipStream_t s = (hipStream_t)stream;
    if (mfr_zero_async(cnt, sizeof(int32_t) * 2 * B, s) != hipSuccess) return MFR_E_LAUNCH;
    hipLaunchKernelGGL(icp_prep_kernel, dim3(nblk, B), dim3(256), 0, s, depth0, depth1, HW, W, K1, k_dtype, R, t, status, Tc, cnt, state);
    CHECK_LAUNCH();
    for (int k = 0; k <= max_iter; ++k) {
        hipLaunchKernelGGL(icp_assoc_kernel, dim3(nblk, B), dim3(256), 0, s, depth0, depth1, Tc, H, W, K0, K1, k_dtype, max_corr_dist, state, partial);
        CHECK_LAUNCH();
        hipLaunchKernelGGL(icp_update_kernel, dim3(B), dim3(64), 0, s, partial, nblk, cnt, k, max_iter, rel_fitness, rel_rmse, state);
        CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(icp_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, state, cnt, status, B, R, t, n_inliers, fitness, rmse, iters);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
