// pnp.hip -- PnP path of the relative-pose solver on gfx950 (MI355X).
//
// Replaces PnPSolver.estimate_pose (lib/models/matching/pose_solver.py:184-235) for a BATCH of
// image pairs, device-resident end to end:
//
//   depth_min_kernel      depth_0.min()                                   (:196, quirk Q6)
//   pnp_lift_kernel       np.int32(pts0) -> depth gather -> valid -> backproject_3d  (:186-206, :6-17)
//   pnp_hyp_score_kernel  cv.solvePnPRansac(P3P) hypotheses + inlier counts (:209-213)
//   pnp_select_kernel     RANSAC best-model replay (adaptive iteration cap), inlier set,
//                         non-minimal refit + ITERATIVE refinement (:216-220), |t|>1000 (:223-225)
//
// RANSAC mapping: hypothesis `it` of pair b is a pure function of (seed, pair_id[b], it)
// (Philox counters), so all max_iters hypotheses are evaluated concurrently -- one LANE per
// minimal solve, then one WAVEFRONT per hypothesis for scoring (lanes stride over the points,
// 64-bit __ballot + popcount gives the inlier count).  OpenCV's sequential early-termination
// rule is then replayed exactly as a prefix-max scan over the counts, so the selected model and
// its inlier set are identical to the sequential CPU loop.
//
// This TU is compiled with -ffp-contract=off (see geom_dev.h for the FP contract).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"
#include "geom_dev.h"

using namespace mfr;

#define MFR_NSEG 16          // depth-min partial segments per image
// pnp_hyp_score_kernel: hypotheses per workgroup / threads per workgroup.  Round 5: 256 / 256 -> 64 / 256 -- a wavefront scored the 64 hypotheses its own
// lanes had solved, one after the other (500 wavefronts on 1024 SIMDs at 1000 iterations x 32 pairs); now wavefront 0 solves a workgroup's 64
// hypotheses and all four wavefronts score 16 each.  A hypothesis' count does not depend on the grouping.
#define HYP_BLOCK 64
#define HYP_THREADS 256
#define PT_TILE 768          // points staged in LDS per tile (24 KB models + 30 KB points < 64 KB)

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) depth_min_kernel(const float *__restrict__ depth, int HW,
                                                        float *__restrict__ partial)
{
    const int b = blockIdx.y, s = blockIdx.x;
    const int seg = (HW + MFR_NSEG - 1) / MFR_NSEG;
    const int lo = s * seg, hi = min(HW, lo + seg);
    const float *d = depth + (size_t)b * HW;
    float m = INFINITY;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) {
        const float v = d[i];
        if (v < m) m = v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(m, off, 64);
        if (o < m) m = o;
    }
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = sm[0];
        for (int w = 1; w < 4; ++w) if (sm[w] < r) r = sm[w];
        partial[b * MFR_NSEG + s] = r;
    }
}

// ------------------------------------------------------------------------------------------
// one workgroup per pair; order-preserving compaction of the valid correspondences
__global__ void __launch_bounds__(256) pnp_lift_kernel(
    const float *__restrict__ pts0, const float *__restrict__ pts1, const int32_t *__restrict__ n_corr, int maxN,
    const float *__restrict__ depth0, const float *__restrict__ partial_min, int H, int W,
    const void *__restrict__ K0, int k_dtype, double *__restrict__ xyz, double *__restrict__ obs,
    int32_t *__restrict__ src_idx, int32_t *__restrict__ n_valid)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    float dmin = partial_min[b * MFR_NSEG];
    for (int s = 1; s < MFR_NSEG; ++s) { const float v = partial_min[b * MFR_NSEG + s]; if (v < dmin) dmin = v; }
    double Ki[4];
    kinv(K0, k_dtype, b, Ki);
    const float *p0 = pts0 + (size_t)b * maxN * 2, *p1 = pts1 + (size_t)b * maxN * 2;
    const float *dm = depth0 + (size_t)b * H * W;
    double *oxyz = xyz + (size_t)b * maxN * 3, *oobs = obs + (size_t)b * maxN * 2;
    int32_t *osrc = src_idx + (size_t)b * maxN;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 256) {
        const int i = start + tid;
        bool valid = false;
        int u = 0, v = 0;
        float d = 0.f;
        if (i < n) {
            u = (int)p0[2 * i]; v = (int)p0[2 * i + 1];                  // np.int32 truncation (:186, Q1)
            if (u >= 0 && u < W && v >= 0 && v < H) {
                d = dm[v * W + u];                                        // :193
                valid = d > dmin;                                         // :196 (Q6)
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
            double X[3];
            backproject(u, v, d, Ki, X);                                  // :206, :6-17
            oxyz[3 * m] = X[0]; oxyz[3 * m + 1] = X[1]; oxyz[3 * m + 2] = X[2];
            oobs[2 * m] = (double)p1[2 * i]; oobs[2 * m + 1] = (double)p1[2 * i + 1];
            osrc[m] = i;
        }
        __syncthreads();
        if (tid == 0) base_s = off + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_valid[b] = base_s;
}

// ------------------------------------------------------------------------------------------
// grid (ceil(iters/256), B).  Phase 1: one lane per hypothesis (sample + P3P + 4th-point
// disambiguation) -> model in LDS.  Phase 2: one wavefront per hypothesis, lanes stride over the
// LDS-staged points, ballot/popcount inlier counting.
__global__ void __launch_bounds__(HYP_THREADS) pnp_hyp_score_kernel(
    const double *__restrict__ xyz, const double *__restrict__ obs, const int32_t *__restrict__ n_valid,
    int maxN, const void *__restrict__ K1, int k_dtype, int max_iters, double thr2, uint64_t seed,
    const int64_t *__restrict__ pair_ids, int32_t *__restrict__ counts)
{
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = n_valid[b];
    const int it = blockIdx.x * HYP_BLOCK + tid;                 // tid < HYP_BLOCK: this thread's hypothesis
    int32_t *cnt_out = counts + (size_t)b * max_iters;
    if (n <= 4) {                      // n < 4: no RANSAC; n == 4: handled by the select kernel
        if (tid < HYP_BLOCK && it < max_iters) cnt_out[it] = -1;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *model = (double *)smem_raw;                       // [12][HYP_BLOCK] (transposed: conflict-free)
    double *px = model + 12 * HYP_BLOCK;                      // SoA point tile
    double *py = px + PT_TILE, *pz = py + PT_TILE, *pu = pz + PT_TILE, *pv = pu + PT_TILE;
    int *mvalid = (int *)(pv + PT_TILE);                      // [HYP_BLOCK]
    int *cnt = mvalid + HYP_BLOCK;                            // [HYP_BLOCK]

    const double *X = xyz + (size_t)b * maxN * 3, *O = obs + (size_t)b * maxN * 2;
    double Kd[4];
    kparams(K1, k_dtype, b, Kd);

    if (tid < HYP_BLOCK) {   // phase 1 (wavefront 0)
        double R[9], t[3];
        int ok = 0;
        if (it < max_iters) {
            int s[4];
            sample_distinct<4>(seed, (uint64_t)pair_ids[b], (uint32_t)it, n, s);
            ok = pnp_hypothesis(X, O, s, Kd, R, t);
        }
        mvalid[tid] = ok;
        cnt[tid] = 0;
        if (ok) {
#pragma unroll
            for (int k = 0; k < 9; ++k) model[k * HYP_BLOCK + tid] = R[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) model[(9 + k) * HYP_BLOCK + tid] = t[k];
        }
    }
    // phase 2
    for (int base = 0; base < n; base += PT_TILE) {
        const int tn = min(PT_TILE, n - base);
        __syncthreads();
        for (int i = tid; i < tn; i += HYP_THREADS) {
            const double *p = X + 3 * (size_t)(base + i);
            const double *o = O + 2 * (size_t)(base + i);
            px[i] = p[0]; py[i] = p[1]; pz[i] = p[2]; pu[i] = o[0]; pv[i] = o[1];
        }
        __syncthreads();
        for (int hi = wid; hi < HYP_BLOCK; hi += HYP_THREADS / 64) {
            if (!mvalid[hi]) continue;                         // wave-uniform
            double R[9], t[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = model[k * HYP_BLOCK + hi];
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = model[(9 + k) * HYP_BLOCK + hi];
            int c = 0;
            for (int i0 = 0; i0 < tn; i0 += 64) {
                const int i = i0 + lane;
                bool in = false;
                if (i < tn) {
                    const double P[3] = { px[i], py[i], pz[i] };
                    const double x2[2] = { pu[i], pv[i] };
                    in = reproj_err2(R, t, P, x2, Kd) <= thr2;
                }
                c += __popcll(__ballot(in));
            }
            if (lane == 0) cnt[hi] += c;
        }
    }
    __syncthreads();
    if (tid < HYP_BLOCK && it < max_iters) cnt_out[it] = mvalid[tid] ? cnt[tid] : 0;
}

// ------------------------------------------------------------------------------------------
// LM pieces (wave-parallel, deterministic wave64 reduction order)
static __device__ __forceinline__ void quat_right_update(const double *R, const double *dw, double *Rn)
{
    const double hx = 0.5 * dw[0], hy = 0.5 * dw[1], hz = 0.5 * dw[2];
    const double nn = sqrt(((hx * hx + hy * hy) + hz * hz) + 1.0);
    const double w = 1.0 / nn, x = hx / nn, y = hy / nn, z = hz / nn;
    double Q[9];
    Q[0] = 1.0 - 2.0 * (y * y + z * z); Q[1] = 2.0 * (x * y - w * z);       Q[2] = 2.0 * (x * z + w * y);
    Q[3] = 2.0 * (x * y + w * z);       Q[4] = 1.0 - 2.0 * (x * x + z * z); Q[5] = 2.0 * (y * z - w * x);
    Q[6] = 2.0 * (x * z - w * y);       Q[7] = 2.0 * (y * z + w * x);       Q[8] = 1.0 - 2.0 * (x * x + y * y);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rn[3 * i + j] = (R[3 * i] * Q[j] + R[3 * i + 1] * Q[3 + j]) + R[3 * i + 2] * Q[6 + j];
}

static __device__ __forceinline__ double pnp_cost(const double *X, const double *O, const int32_t *idx, int n,
                                                  const double *Kd, const double *R, const double *t)
{
    double acc = 0.0;
    for (int i = lane_id(); i < n; i += 64) {
        const int j = idx[i];
        acc = acc + reproj_err2(R, t, X + 3 * (size_t)j, O + 2 * (size_t)j, Kd);
    }
    return wave_sum(acc);
}

static __device__ __forceinline__ int chol_solve6(const double *A, const double *bvec, double *x)
{
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
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
        double s = bvec[i];
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

// Levenberg-Marquardt on the reprojection error over the inlier list; stands in for the EPnP
// refit inside cv::solvePnPRansac and for cv.solvePnPGeneric(ITERATIVE) (pose_solver.py:216-220).
static __device__ __noinline__ int pnp_lm(const double *X, const double *O, const int32_t *idx, int n_idx,
                                          const double *Kd, int max_iter, double *R, double *t)
{
    double lambda = 1e-3;
    double cost = pnp_cost(X, O, idx, n_idx, Kd, R, t);
    if (!(cost == cost) || !(cost < 1e300)) return -1;
    for (int it = 0; it < max_iter; ++it) {
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        for (int i = lane_id(); i < n_idx; i += 64) {
            const int j = idx[i];
            const double *P = X + 3 * (size_t)j, *x = O + 2 * (size_t)j;
            double Y[3];
            rot_apply(R, t, P, Y);
            const double iz = (Y[2] != 0.0) ? 1.0 / Y[2] : 1.0;
            const double xn = Y[0] * iz, yn = Y[1] * iz;
            const double ru = (Kd[0] * xn + Kd[2]) - x[0];
            const double rv = (Kd[1] * yn + Kd[3]) - x[1];
            const double a0[3] = { 0.0, -P[2], P[1] }, a1[3] = { P[2], 0.0, -P[0] }, a2[3] = { -P[1], P[0], 0.0 };
            double G[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                G[r][0] = (R[3 * r] * a0[0] + R[3 * r + 1] * a0[1]) + R[3 * r + 2] * a0[2];
                G[r][1] = (R[3 * r] * a1[0] + R[3 * r + 1] * a1[1]) + R[3 * r + 2] * a1[2];
                G[r][2] = (R[3 * r] * a2[0] + R[3 * r + 1] * a2[1]) + R[3 * r + 2] * a2[2];
                G[r][3] = (r == 0) ? 1.0 : 0.0; G[r][4] = (r == 1) ? 1.0 : 0.0; G[r][5] = (r == 2) ? 1.0 : 0.0;
            }
            const double pu0 = Kd[0] * iz, pu2 = -(Kd[0] * xn) * iz;
            const double pv1 = Kd[1] * iz, pv2 = -(Kd[1] * yn) * iz;
            double Ju[6], Jv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                Ju[k] = pu0 * G[0][k] + pu2 * G[2][k];
                Jv[k] = pv1 * G[1][k] + pv2 * G[2][k];
            }
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = r; c < 6; ++c, ++q) acc[q] = acc[q] + (Ju[r] * Ju[c] + Jv[r] * Jv[c]);
#pragma unroll
            for (int r = 0; r < 6; ++r, ++q) acc[q] = acc[q] + (Ju[r] * ru + Jv[r] * rv);
        }
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = wave_sum(acc[q]);
        double Hm[36], g[6];
        {
            int q = 0;
            for (int r = 0; r < 6; ++r)
                for (int c = r; c < 6; ++c, ++q) { Hm[6 * r + c] = acc[q]; Hm[6 * c + r] = acc[q]; }
            for (int r = 0; r < 6; ++r, ++q) g[r] = -acc[q];
        }
        for (int r = 0; r < 6; ++r) Hm[6 * r + r] = Hm[6 * r + r] + lambda * Hm[6 * r + r];
        double dlt[6];
        if (chol_solve6(Hm, g, dlt)) {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
            continue;
        }
        double Rn[9], tn[3];
        quat_right_update(R, dlt, Rn);
        tn[0] = t[0] + dlt[3]; tn[1] = t[1] + dlt[4]; tn[2] = t[2] + dlt[5];
        const double cn = pnp_cost(X, O, idx, n_idx, Kd, Rn, tn);
        double mx = 0.0;
        for (int k = 0; k < 6; ++k) { const double a = dlt[k] < 0.0 ? -dlt[k] : dlt[k]; if (a > mx) mx = a; }
        if (cn < cost) {
            const double dec = cost - cn;
            for (int k = 0; k < 9; ++k) R[k] = Rn[k];
            for (int k = 0; k < 3; ++k) t[k] = tn[k];
            const bool done = (dec <= 1e-14 * cost);
            cost = cn;
            lambda = lambda * 0.1;
            if (lambda < 1e-12) lambda = 1e-12;
            if (done) break;
        } else {
            lambda = lambda * 10.0;
            if (lambda > 1e12) break;
        }
        if (mx < 1e-13) break;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// one wavefront per pair: replay of RANSACPointSetRegistrator::run's best-model / adaptive
// iteration-cap logic over the precomputed counts, inlier set, refit + refinement, checks.
__global__ void __launch_bounds__(64) pnp_select_kernel(
    const double *__restrict__ xyz, const double *__restrict__ obs, const int32_t *__restrict__ n_valid,
    const int32_t *__restrict__ pre_status, int maxN, const void *__restrict__ K1, int k_dtype, int max_iters,
    double thr2, double conf, uint64_t seed, const int64_t *__restrict__ pair_ids,
    const int32_t *__restrict__ counts, int32_t *__restrict__ inl_idx,
    double *__restrict__ Rout, double *__restrict__ tout, int32_t *__restrict__ n_inliers,
    int32_t *__restrict__ status, uint8_t *__restrict__ mask_valid, int32_t *__restrict__ best_iter,
    int32_t *__restrict__ iters_run)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = n_valid[b];
    const double *X = xyz + (size_t)b * maxN * 3, *O = obs + (size_t)b * maxN * 2;
    double Kd[4];
    kparams(K1, k_dtype, b, Kd);
    int32_t *idx = inl_idx + (size_t)b * maxN;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);

    int st = pre_status ? pre_status[b] : MFR_ST_OK;
    if (st == MFR_ST_OK && n < 4) st = MFR_ST_TOO_FEW;
    int bit = -1, best = 3, run = 0, m = 0;
    double R[9], t[3];
    if (mask_valid)
        for (int i = lane; i < maxN; i += 64) mask_valid[(size_t)b * maxN + i] = 0;

    if (st == MFR_ST_OK) {
        if (n == 4) {
            const int s[4] = { 0, 1, 2, 3 };
            if (pnp_hypothesis(X, O, s, Kd, R, t)) { bit = 0; best = 4; run = 1; } else st = MFR_ST_NO_MODEL;
        } else {
            // records of the running max, then sequential replay of the iteration cap
            int niters = max_iters, carry = 3;
            bool stop = false;
            const int32_t *cnt = counts + (size_t)b * max_iters;
            for (int c0 = 0; c0 < max_iters && !stop && c0 < niters; c0 += 64) {
                const int it = c0 + lane;
                const int v = (it < max_iters) ? cnt[it] : -1;
                int incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off, 64);
                    if (lane >= off && o > incl) incl = o;
                }
                int excl = __shfl_up(incl, 1, 64);
                if (lane == 0 || excl < carry) excl = carry;
                unsigned long long rec = __ballot(v > excl);
                while (rec) {
                    const int l = __ffsll((long long)rec) - 1;
                    rec &= rec - 1;
                    const int itr = c0 + l;
                    if (itr >= niters) { stop = true; break; }
                    best = __shfl(v, l, 64);
                    bit = itr;
                    niters = update_num_iters(conf, (double)(n - best) / (double)n, 4, niters);
                }
                const int last = __shfl(incl, 63, 64);
                if (last > carry) carry = last;
            }
            run = (bit + 1 > niters) ? bit + 1 : niters;   // loop exit index of the sequential form
            if (bit < 0) st = MFR_ST_NO_MODEL;
            else {
                int s[4];
                sample_distinct<4>(seed, (uint64_t)pair_ids[b], (uint32_t)bit, n, s);
                if (!pnp_hypothesis(X, O, s, Kd, R, t)) st = MFR_ST_NO_MODEL;   // cannot happen (count > 3)
            }
        }
    }
    if (st == MFR_ST_OK) {
        // inlier list of the best model (ascending index order)
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            bool in = false;
            if (i < n) in = (n == 4) ? true : (reproj_err2(R, t, X + 3 * (size_t)i, O + 2 * (size_t)i, Kd) <= thr2);
            const unsigned long long bal = __ballot(in);
            if (in) idx[m + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            if (mask_valid && i < n) mask_valid[(size_t)b * maxN + i] = in ? 1 : 0;
            m += __popcll(bal);
        }
        __threadfence();          // idx[] written by other lanes of this wave is read below
        if (n > 4) {
            if (pnp_lm(X, O, idx, m, Kd, 20, R, t)) st = MFR_ST_NO_MODEL;
            if (st == MFR_ST_OK && m >= 6)
                if (pnp_lm(X, O, idx, m, Kd, 20, R, t)) st = MFR_ST_NO_MODEL;   // pose_solver.py:216-220
        }
    }
    if (st == MFR_ST_OK) {
        bool bad = false;
        for (int k = 0; k < 9; ++k) bad |= !(R[k] == R[k]);
        for (int k = 0; k < 3; ++k) bad |= !(t[k] == t[k]);
        if (bad) st = MFR_ST_NO_MODEL;
    }
    if (st == MFR_ST_OK) {
        const double tn = sqrt(dot3(t, t));
        if (tn > 1000.0) st = MFR_ST_DEGENERATE;                                 // pose_solver.py:223-225
    }
    if (lane == 0) {
        for (int k = 0; k < 9; ++k) Rout[9 * b + k] = (st == MFR_ST_OK) ? R[k] : qnan;
        for (int k = 0; k < 3; ++k) tout[3 * b + k] = (st == MFR_ST_OK) ? t[k] : qnan;
        n_inliers[b] = (st == MFR_ST_OK) ? m : 0;
        status[b] = st;
        if (best_iter) best_iter[b] = bit;
        if (iters_run) iters_run[b] = run;
    }
}

// pre-status: too-few / bad-depth decided from counts (pose_solver.py:188-189,197-198)
__global__ void pnp_prestatus_kernel(const int32_t *n_corr, const int32_t *n_valid, int B, int32_t *pre)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int st = MFR_ST_OK;
    if (n_corr[b] < 4) st = MFR_ST_TOO_FEW;
    else if (n_valid[b] < 4) st = MFR_ST_BAD_DEPTH;
    pre[b] = st;
}

// scatter the inlier mask over lifted points back to original correspondence indices
__global__ void pnp_mask_scatter_kernel(const uint8_t *mask_valid, const int32_t *src_idx, const int32_t *n_valid,
                                        const int32_t *status, int maxN, uint8_t *mask_out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= maxN) return;
    // mask_out was zeroed by the caller-side memset node
    if (status[b] == MFR_ST_OK && i < n_valid[b] && mask_valid[(size_t)b * maxN + i])
        mask_out[(size_t)b * maxN + src_idx[(size_t)b * maxN + i]] = 1;
}

// ------------------------------------------------------------------------------------------
// test hooks
__global__ void f64_ops_kernel(const double *a, const double *b, const double *c, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * i] = a[i] / b[i];
    out[3 * i + 1] = sqrt(a[i] < 0 ? -a[i] : a[i]);
    out[3 * i + 2] = a[i] * b[i] + c[i];
}
template <int K>
__global__ void sample_kernel(uint64_t seed, const int64_t *pair_ids, int iters, int n, int32_t *out)
{
    const int b = blockIdx.y;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= iters) return;
    int s[K];
    sample_distinct<K>(seed, (uint64_t)pair_ids[b], (uint32_t)it, n, s);
    for (int k = 0; k < K; ++k) out[((size_t)b * iters + it) * K + k] = s[k];
}

// ------------------------------------------------------------------------------------------
// C-ABI
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

extern "C" {

int mfr_abi_version(void) { return MFR_ABI_VERSION; }
const char *mfr_target_arch(void) { return "gfx950"; }

int mfr_test_f64_ops(const double *a, const double *b, const double *c, int n, double *out3, void *stream)
{
    if (!a || !b || !c || !out3 || n < 0) return MFR_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(f64_ops_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, c, n, out3);
    CHECK_LAUNCH();
    return 0;
}

int mfr_test_sample(uint64_t seed, const int64_t *pair_ids, int B, int iters, int n, int k, int32_t *out, void *stream)
{
    if (!pair_ids || !out || B <= 0 || iters <= 0 || n < k || (k != 4 && k != 5)) return MFR_E_ARG;
    dim3 grid((iters + 255) / 256, B);
    if (k == 4) hipLaunchKernelGGL(sample_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, seed, pair_ids, iters, n, out);
    else hipLaunchKernelGGL(sample_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, seed, pair_ids, iters, n, out);
    CHECK_LAUNCH();
    return 0;
}

int mfr_depth_min(const float *depth, int B, int H, int W, float *partial_min, void *stream)
{
    if (!depth || !partial_min || B <= 0 || H <= 0 || W <= 0) return MFR_E_ARG;
    hipLaunchKernelGGL(depth_min_kernel, dim3(MFR_NSEG, B), dim3(256), 0, (hipStream_t)stream, depth, H * W, partial_min);
    CHECK_LAUNCH();
    return 0;
}

int mfr_pnp_lift(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                 const float *depth0, const float *partial_min, int H, int W, const void *K0, int k_dtype,
                 double *xyz, double *obs, int32_t *src_idx, int32_t *n_valid, void *stream)
{
    if (!pts0 || !pts1 || !n_corr || !depth0 || !partial_min || !K0 || !xyz || !obs || !src_idx || !n_valid ||
        B <= 0 || maxN <= 0 || H <= 0 || W <= 0 || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    hipLaunchKernelGGL(pnp_lift_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pts0, pts1, n_corr, maxN,
                       depth0, partial_min, H, W, K0, k_dtype, xyz, obs, src_idx, n_valid);
    CHECK_LAUNCH();
    return 0;
}

static size_t hyp_smem_bytes(void)
{
    return (size_t)(12 * HYP_BLOCK + 5 * PT_TILE) * sizeof(double) + 2 * HYP_BLOCK * sizeof(int);
}

static int launch_ransac(const double *xyz, const double *obs, const int32_t *n_valid, const int32_t *pre_status,
                         int B, int maxN, const void *K1, int k_dtype, int max_iters, double thr, double conf, uint64_t seed,
                         const int64_t *pair_ids, int32_t *counts, int32_t *inl_idx, double *R, double *t,
                         int32_t *n_inliers, int32_t *status, uint8_t *mask_valid, int32_t *best_iter,
                         int32_t *iters_run, hipStream_t s)
{
    const double thr2 = thr * thr;
    hipLaunchKernelGGL(pnp_hyp_score_kernel, dim3((max_iters + HYP_BLOCK - 1) / HYP_BLOCK, B), dim3(HYP_THREADS),
                       hyp_smem_bytes(), s, xyz, obs, n_valid, maxN, K1, k_dtype, max_iters, thr2, seed, pair_ids, counts);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(pnp_select_kernel, dim3(B), dim3(64), 0, s, xyz, obs, n_valid, pre_status, maxN, K1, k_dtype, max_iters,
                       thr2, conf, seed, pair_ids, counts, inl_idx, R, t, n_inliers, status, mask_valid, best_iter,
                       iters_run);
    CHECK_LAUNCH();
    return 0;
}

int mfr_pnp_ransac(const double *xyz, const double *obs, const int32_t *n_valid, int B, int maxN,
                   const void *K1, int k_dtype, int max_iters, double reproj_thr, double confidence,
                   uint64_t seed, const int64_t *pair_ids, int32_t *counts, int32_t *inl_idx,
                   double *R, double *t, int32_t *n_inliers, int32_t *status, uint8_t *mask_valid,
                   int32_t *best_iter, int32_t *iters_run, void *stream)
{
    if (!xyz || !obs || !n_valid || !K1 || !pair_ids || !counts || !inl_idx || !R || !t || !n_inliers || !status ||
        B <= 0 || maxN <= 0 || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    if (max_iters < 1) max_iters = 1;
    return launch_ransac(xyz, obs, n_valid, nullptr, B, maxN, K1, k_dtype, max_iters, reproj_thr, confidence, seed, pair_ids,
                         counts, inl_idx, R, t, n_inliers, status, mask_valid, best_iter, iters_run, (hipStream_t)stream);
}

// workspace layout of mfr_pnp_solve_batch (all 256-B aligned)
struct PnpWs { size_t partial, xyz, obs, src, nvalid, pre, counts, inl, maskv, total; };
static PnpWs pnp_ws_layout(int B, int maxN, int max_iters)
{
    PnpWs w; size_t o = 0;
    w.partial = o; o = align_up(o + sizeof(float) * MFR_NSEG * (size_t)B, 256);
    w.xyz = o;     o = align_up(o + sizeof(double) * 3 * (size_t)B * maxN, 256);
    w.obs = o;     o = align_up(o + sizeof(double) * 2 * (size_t)B * maxN, 256);
    w.src = o;     o = align_up(o + sizeof(int32_t) * (size_t)B * maxN, 256);
    w.nvalid = o;  o = align_up(o + sizeof(int32_t) * (size_t)B, 256);
    w.pre = o;     o = align_up(o + sizeof(int32_t) * (size_t)B, 256);
    w.counts = o;  o = align_up(o + sizeof(int32_t) * (size_t)B * max_iters, 256);
    w.inl = o;     o = align_up(o + sizeof(int32_t) * (size_t)B * maxN, 256);
    w.maskv = o;   o = align_up(o + (size_t)B * maxN, 256);
    w.total = o;
    return w;
}

size_t mfr_pnp_workspace_bytes(int B, int maxN, int max_iters)
{
    if (B <= 0 || maxN <= 0) return 0;
    if (max_iters < 1) max_iters = 1;
    return pnp_ws_layout(B, maxN, max_iters).total;
}

int mfr_pnp_solve_batch(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                        const float *depth0, int H, int W, const void *K0, const void *K1, int k_dtype,
                        int max_iters, double reproj_thr, double confidence, uint64_t seed, const int64_t *pair_ids,
                        void *workspace, size_t workspace_bytes,
                        double *R, double *t, int32_t *n_inliers, int32_t *status, uint8_t *inlier_mask, void *stream)
{
    if (!pts0 || !pts1 || !n_corr || !depth0 || !K0 || !K1 || !pair_ids || !workspace || !R || !t || !n_inliers ||
        !status || B <= 0 || maxN <= 0 || H <= 0 || W <= 0 || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    if (max_iters < 1) max_iters = 1;
    const PnpWs w = pnp_ws_layout(B, maxN, max_iters);
    if (workspace_bytes < w.total) return MFR_E_WORKSPACE;
    char *ws = (char *)workspace;
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)(ws + w.partial);
    double *xyz = (double *)(ws + w.xyz), *obs = (double *)(ws + w.obs);
    int32_t *src = (int32_t *)(ws + w.src), *nvalid = (int32_t *)(ws + w.nvalid), *pre = (int32_t *)(ws + w.pre);
    int32_t *counts = (int32_t *)(ws + w.counts), *inl = (int32_t *)(ws + w.inl);
    uint8_t *maskv = (uint8_t *)(ws + w.maskv);

    int rc = mfr_depth_min(depth0, B, H, W, partial, stream);
    if (rc) return rc;
    rc = mfr_pnp_lift(pts0, pts1, n_corr, B, maxN, depth0, partial, H, W, K0, k_dtype, xyz, obs, src, nvalid, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(pnp_prestatus_kernel, dim3((B + 63) / 64), dim3(64), 0, s, n_corr, nvalid, B, pre);
    CHECK_LAUNCH();
    rc = launch_ransac(xyz, obs, nvalid, pre, B, maxN, K1, k_dtype, max_iters, reproj_thr, confidence, seed, pair_ids, counts,
                       inl, R, t, n_inliers, status, inlier_mask ? maskv : nullptr, nullptr, nullptr, s);
    if (rc) return rc;
    if (inlier_mask) {
        if (mfr_zero_async(inlier_mask, (size_t)B * maxN, s) != hipSuccess) return MFR_E_LAUNCH;
        hipLaunchKernelGGL(pnp_mask_scatter_kernel, dim3((maxN + 255) / 256, B), dim3(256), 0, s, maskv, src, nvalid,
                           status, maxN, inlier_mask);
        CHECK_LAUNCH();
    }
    return 0;
}

}  // extern "C"
