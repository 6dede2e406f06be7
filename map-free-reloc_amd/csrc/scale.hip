// scale.hip -- metric scale from depth on gfx950: the part of
// EssentialMatrixMetricSolver.estimate_pose the reference itself owns
// (lib/models/matching/pose_solver.py:137-172), batched over image pairs.
//
//   scale_lift_kernel    mask==1 (:137) -> np.int32 truncation (:138-139) -> depth gather (:140-141)
//                        -> valid d0>0 & d1>0 (:144) -> backproject (:150-151) -> R xyz0 (:154)
//                        -> scale_i = (xyz1_i - R xyz0_i) . t (:157), order-preserving compaction
//   scale_ransac_kernel  exhaustive 1-D RANSAC: hypothesis i = scale_i, count_j |s_j - s_i| < thr
//                        (:160-166).  One LANE per hypothesis, scales staged in LDS and read as
//                        wave-wide broadcasts; per-workgroup argmax (count desc, index asc).
//   scale_final_kernel   first strict maximum over the workgroup partials (Q2), t_metric = s * t (:169)
//
// The reference's O(N^2) python loop is the whole cost of this stage on CPU; here it is
// N^2 compare-adds on LDS-resident data (8N bytes of input per pair).
// Compiled with -ffp-contract=off (FP contract in geom_dev.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "geom_dev.h"

using namespace mfr;

#define SC_BLOCK 256
#define SC_TILE 2048

__global__ void __launch_bounds__(256) scale_lift_kernel(
    const float *__restrict__ pts0, const float *__restrict__ pts1, const uint8_t *__restrict__ emat_mask,
    const int32_t *__restrict__ n_corr, int maxN, const float *__restrict__ depth0,
    const float *__restrict__ depth1, int H, int W, const void *__restrict__ K0, const void *__restrict__ K1, int k_dtype,
    const double *__restrict__ Rin, const double *__restrict__ tin, const int32_t *__restrict__ in_status,
    double *__restrict__ scale, int32_t *__restrict__ n_scale)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int n = n_corr[b];
    if (n > maxN) n = maxN;
    if (in_status && in_status[b] != MFR_ST_OK) n = 0;
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    double Ki0[4], Ki1[4];
    kinv(K0, k_dtype, b, Ki0);
    kinv(K1, k_dtype, b, Ki1);
    double R[9], t[3];
    for (int k = 0; k < 9; ++k) R[k] = Rin[9 * b + k];
    for (int k = 0; k < 3; ++k) t[k] = tin[3 * b + k];
    const float *p0 = pts0 + (size_t)b * maxN * 2, *p1 = pts1 + (size_t)b * maxN * 2;
    const uint8_t *mk = emat_mask ? emat_mask + (size_t)b * maxN : nullptr;
    const float *d0m = depth0 + (size_t)b * H * W, *d1m = depth1 + (size_t)b * H * W;
    double *out = scale + (size_t)b * maxN;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 256) {
        const int i = start + tid;
        bool valid = false;
        double sc = 0.0;
        if (i < n && (!mk || mk[i] == 1)) {
            const int u0 = (int)p0[2 * i], v0 = (int)p0[2 * i + 1];
            const int u1 = (int)p1[2 * i], v1 = (int)p1[2 * i + 1];
            if (u0 >= 0 && u0 < W && v0 >= 0 && v0 < H && u1 >= 0 && u1 < W && v1 >= 0 && v1 < H) {
                const float d0 = d0m[v0 * W + u0], d1 = d1m[v1 * W + u1];
                if (d0 > 0.f && d1 > 0.f) {
                    valid = true;
                    double a[3], c[3], ra[3];
                    backproject(u0, v0, d0, Ki0, a);
                    backproject(u1, v1, d1, Ki1, c);
                    ra[0] = (R[0] * a[0] + R[1] * a[1]) + R[2] * a[2];
                    ra[1] = (R[3] * a[0] + R[4] * a[1]) + R[5] * a[2];
                    ra[2] = (R[6] * a[0] + R[7] * a[1]) + R[8] * a[2];
                    const double d[3] = { c[0] - ra[0], c[1] - ra[1], c[2] - ra[2] };
                    sc = dot3(d, t);
                }
            }
        }
        const unsigned long long bal = __ballot(valid);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wid; ++w) off += wave_cnt[w];
        if (valid) out[off + wpre] = sc;
        __syncthreads();
        if (tid == 0) base_s = off + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_scale[b] = base_s;
}

// grid (chunks, B): workgroup c owns hypotheses [c*256, c*256+256)
__global__ void __launch_bounds__(SC_BLOCK) scale_ransac_kernel(
    const double *__restrict__ scale, const int32_t *__restrict__ n_scale, int maxN, double thr,
    int32_t *__restrict__ part_cnt, int32_t *__restrict__ part_idx, int nchunks)
{
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = n_scale[b];
    if (c * SC_BLOCK >= n) {
        if (tid == 0) { part_cnt[b * nchunks + c] = 0; part_idx[b * nchunks + c] = -1; }
        return;
    }
    __shared__ double tile[SC_TILE];
    __shared__ int red_cnt[4], red_idx[4];
    const double *s = scale + (size_t)b * maxN;
    const int hi = c * SC_BLOCK + tid;
    const double sh = (hi < n) ? s[hi] : 0.0;
    int cnt = 0;
    for (int base = 0; base < n; base += SC_TILE) {
        const int tn = min(SC_TILE, n - base);
        __syncthreads();
        for (int i = tid; i < tn; i += SC_BLOCK) tile[i] = s[base + i];
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
            double d = tile[j] - sh;
            if (d < 0.0) d = -d;
            cnt += (d < thr);
        }
    }
    if (hi >= n) cnt = -1;
    // argmax: larger count wins, ties -> smaller index (first strict maximum, :162-166)
    int bc = cnt, bi = hi;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int oc = __shfl_xor(bc, off, 64), oi = __shfl_xor(bi, off, 64);
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (lane == 0) { red_cnt[wid] = bc; red_idx[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red_cnt[w] > bc || (red_cnt[w] == bc && red_idx[w] < bi)) { bc = red_cnt[w]; bi = red_idx[w]; }
        part_cnt[b * nchunks + c] = bc;
        part_idx[b * nchunks + c] = bi;
    }
}

__global__ void scale_final_kernel(const double *__restrict__ scale, const int32_t *__restrict__ n_scale, int maxN,
                                   const int32_t *__restrict__ part_cnt, const int32_t *__restrict__ part_idx,
                                   int nchunks, const double *__restrict__ tin, const int32_t *__restrict__ in_status,
                                   int B, double *__restrict__ t_metric, double *__restrict__ best_scale,
                                   int32_t *__restrict__ n_inliers, int32_t *__restrict__ status)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    int st = in_status ? in_status[b] : MFR_ST_OK;
    int bc = 0, bi = -1;
    if (st == MFR_ST_OK) {
        if (n_scale[b] < 1) st = MFR_ST_BAD_DEPTH;                       // :145-149
        else {
            for (int c = 0; c < nchunks; ++c) {
                const int pc = part_cnt[b * nchunks + c], pi = part_idx[b * nchunks + c];
                if (pi >= 0 && (pc > bc || (pc == bc && bi >= 0 && pi < bi))) { bc = pc; bi = pi; }
            }
            if (bi < 0) st = MFR_ST_NO_MODEL;
        }
    }
    if (st == MFR_ST_OK) {
        const double sc = scale[(size_t)b * maxN + bi];
        best_scale[b] = sc;
        for (int k = 0; k < 3; ++k) t_metric[3 * b + k] = sc * tin[3 * b + k];   // :169
        n_inliers[b] = bc;
    } else {
        best_scale[b] = qnan;
        for (int k = 0; k < 3; ++k) t_metric[3 * b + k] = qnan;
        n_inliers[b] = 0;
    }
    status[b] = st;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

struct ScaleWs { size_t scale, nscale, pcnt, pidx, total; int nchunks; };
static ScaleWs scale_ws_layout(int B, int maxN)
{
    ScaleWs w; size_t o = 0;
    w.nchunks = (maxN + SC_BLOCK - 1) / SC_BLOCK;
    w.scale = o;  o = align_up(o + sizeof(double) * (size_t)B * maxN, 256);
    w.nscale = o; o = align_up(o + sizeof(int32_t) * (size_t)B, 256);
    w.pcnt = o;   o = align_up(o + sizeof(int32_t) * (size_t)B * w.nchunks, 256);
    w.pidx = o;   o = align_up(o + sizeof(int32_t) * (size_t)B * w.nchunks, 256);
    w.total = o;
    return w;
}

extern "C" {

size_t mfr_scale_workspace_bytes(int B, int maxN)
{
    if (B <= 0 || maxN <= 0) return 0;
    return scale_ws_layout(B, maxN).total;
}

int mfr_scale_from_depth_batch(const float *pts0, const float *pts1, const uint8_t *emat_mask,
                               const int32_t *n_corr, int B, int maxN,
                               const float *depth0, const float *depth1, int H, int W,
                               const void *K0, const void *K1, int k_dtype, const double *R, const double *t,
                               const int32_t *in_status, double scale_thr,
                               void *workspace, size_t workspace_bytes,
                               double *t_metric, double *best_scale, int32_t *n_inliers, int32_t *status, void *stream)
{
    if (!pts0 || !pts1 || !n_corr || !depth0 || !depth1 || !K0 || !K1 || !R || !t || !workspace || !t_metric ||
        !best_scale || !n_inliers || !status || B <= 0 || maxN <= 0 || H <= 0 || W <= 0 || !k_dtype_ok(k_dtype)) return MFR_E_ARG;
    const ScaleWs w = scale_ws_layout(B, maxN);
    if (workspace_bytes < w.total) return MFR_E_WORKSPACE;
    char *ws = (char *)workspace;
    hipStream_t s = (hipStream_t)stream;
    double *scale = (double *)(ws + w.scale);
    int32_t *nscale = (int32_t *)(ws + w.nscale), *pcnt = (int32_t *)(ws + w.pcnt), *pidx = (int32_t *)(ws + w.pidx);
    hipLaunchKernelGGL(scale_lift_kernel, dim3(B), dim3(256), 0, s, pts0, pts1, emat_mask, n_corr, maxN, depth0,
                       depth1, H, W, K0, K1, k_dtype, R, t, in_status, scale, nscale);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(scale_ransac_kernel, dim3(w.nchunks, B), dim3(SC_BLOCK), 0, s, scale, nscale, maxN, scale_thr,
                       pcnt, pidx, w.nchunks);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(scale_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, scale, nscale, maxN, pcnt, pidx,
                       w.nchunks, t, in_status, B, t_metric, best_scale, n_inliers, status);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
