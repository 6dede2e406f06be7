// descriptor_match.hip -- rootSIFT normalisation + exhaustive 2-nearest-neighbour search + Lowe ratio
// test over 128-d descriptors on gfx950 (hot-path row a-3).
//
// Reference call sites: SIFTMatching.get_correspondences (lib/models/matching/feature_matching.py:75-118)
// and SIFT_matcher.match (etc/feature_matching_baselines/matchers.py:135-188):
//     des = root_sift(des)                                  (L1-normalise, sqrt; :68-74 / :127-133)
//     matches = FlannBasedMatcher(kd-tree 5 trees, 50 checks).knnMatch(des0, des1, k=2)
//     keep m where m.distance < ratio * n.distance          (:97-101 / :170-174), in query order
// FLANN's randomised kd-forest is an APPROXIMATE 2-NN; this kernel is the exact search it approximates
// (2048 x 2048 x 128 is 1 GFLOP per pair -- nothing on the matrix cores), so its output is what FLANN
// returns whenever FLANN's search succeeds.  SIFT detection/description itself (cv.SIFT_create) stays
// outside: the kernels start from the [N,128] descriptor matrices detectAndCompute returns.
//
// Mapping to CDNA4 (same skeleton as the SuperGlue attention kernel): one wavefront owns 32 query
// descriptors, held for the whole sweep as the B operand of v_mfma_f32_32x32x2_f32 (64 VGPRs per lane);
// the train descriptors stream through LDS in 32-row tiles (row stride 132 floats: conflict-free 16-B
// reads), register-prefetched and double-buffered; d^2 = |a|^2 + |b|^2 - 2 a.b with the dot product on
// the exact-fp32 matrix cores; the running (best, second-best) pair lives in registers per lane and the
// two lane halves are merged once at the end.  HBM traffic is the two descriptor matrices once (L2
// serves the re-reads across query blocks: (pair) is the fastest grid index -> same XCD).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DM_D 128
#define DM_KT 32
#define DM_KS 132
#define DM_QW 32
#define DM_WAVES 4

// correctly rounded f32 divide and square root, evaluated through binary64 (53 >= 2*24+2 bits, so the
// second rounding is innocuous): identical to numpy's f32 `/` and np.sqrt whatever the fast-math mode
__device__ __forceinline__ float sqrt_rn_f32(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float rs_elem(float a, float den) { return sqrt_rn_f32((float)((double)a / (double)den)); }

// root_sift (feature_matching.py:68-74): d / (sum(d) + 1e-7f) then sqrt, all in f32; the row sum in
// numpy's pairwise order for a 128-element contiguous row (8 strided accumulators, then a 3-level
// tree), so the result is bit-identical to the reference's numpy for any input.  Also emits |y|^2.
__global__ void __launch_bounds__(256) rootsift_kernel(const float *__restrict__ in, int n_rows,
                                                       float *__restrict__ out, float *__restrict__ norm2)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const float4 *p = (const float4 *)(in + (size_t)row * DM_D);
    float r[8];
    {
        const float4 a = p[0], b = p[1];
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    }
    for (int i = 1; i < 16; ++i) {
        const float4 a = p[2 * i], b = p[2 * i + 1];
        r[0] += a.x; r[1] += a.y; r[2] += a.z; r[3] += a.w; r[4] += b.x; r[5] += b.y; r[6] += b.z; r[7] += b.w;
    }
    const float sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const float den = sum + 1e-7f;
    float4 *o = (float4 *)(out + (size_t)row * DM_D);
    float n2 = 0.f;
    for (int i = 0; i < 32; ++i) {
        const float4 a = p[i];
        float4 y;
        y.x = rs_elem(a.x, den); y.y = rs_elem(a.y, den); y.z = rs_elem(a.z, den); y.w = rs_elem(a.w, den);
        n2 = fmaf(y.x, y.x, n2); n2 = fmaf(y.y, y.y, n2); n2 = fmaf(y.z, y.z, n2); n2 = fmaf(y.w, y.w, n2);
        o[i] = y;
    }
    norm2[row] = n2;
}

// exact 2-NN of every query descriptor among the pair's train descriptors (squared L2, ties -> lower index)
__global__ void __launch_bounds__(256, 2) desc_2nn_kernel(
    const float *__restrict__ des0, const float *__restrict__ des1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, int B, int N0, int N1, const int *__restrict__ n0v, const int *__restrict__ n1v,
    int *__restrict__ nn_idx, float *__restrict__ nn_d2)
{
    __shared__ __attribute__((aligned(16))) float Ks[2][DM_KT][DM_KS];
    __shared__ float Kn[2][DM_KT];
    const int b = blockIdx.x % B, qb = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ql = lane & 31, half = lane >> 5;
    const int nq = n0v[b], nk = n1v[b];
    const int q0 = qb * (DM_QW * DM_WAVES);
    const int q = q0 + wid * DM_QW + ql;
    if (q0 >= nq) return;                                   // rows >= n0 are not defined

    float qreg[64];
    const bool qok = q < nq;
    {
        const float4 *qp = (const float4 *)(des0 + ((size_t)b * N0 + (qok ? q : 0)) * DM_D + half * 64);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4 t = qp[g];
            qreg[4 * g] = t.x; qreg[4 * g + 1] = t.y; qreg[4 * g + 2] = t.z; qreg[4 * g + 3] = t.w;
        }
    }
    const float qn = nrm0[(size_t)b * N0 + (qok ? q : 0)];

    // staging: 256 threads x 4 float4 = 32 rows x 128 floats; thread -> rows sr + 8*i, column sc
    const int sr = tid >> 5, sc = (tid & 31) * 4;
    const float *kbase = des1 + (size_t)b * N1 * DM_D + sc;
    const int ntiles = (nk + DM_KT - 1) / DM_KT;
    float4 kr[4];
    float knr = 0.f;
    auto gload = [&](int t) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = t * DM_KT + sr + 8 * i;
            kr[i] = (k < nk) ? *(const float4 *)(kbase + (size_t)k * DM_D) : z;
        }
        if (tid < DM_KT) { const int k = t * DM_KT + tid; knr = (k < nk) ? nrm1[(size_t)b * N1 + k] : 0.f; }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(float4 *)&Ks[buf][sr + 8 * i][sc] = kr[i];
        if (tid < DM_KT) Kn[buf][tid] = knr;
    };
    if (ntiles > 0) { gload(0); lstore(0); }
    __syncthreads();

    float b1 = INFINITY, b2 = INFINITY;
    int i1 = -1;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float *krow = &Ks[buf][ql][half * 64];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4 a = *(const float4 *)(krow + 4 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * g], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * g + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * g + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * g + 3], s, 0, 0, 0);
        }
        // accumulator row (r, half) = train row (r&3) + 8(r>>2) + 4 half of this tile, column = query ql
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = 4 * half + (r & 3) + 8 * (r >> 2);
            const int key = t * DM_KT + kl;
            const float d2 = fmaxf((qn + Kn[buf][kl]) - 2.f * s[r], 0.f);
            if (key < nk) {
                if (d2 < b1 || (d2 == b1 && key < i1)) { b2 = b1; b1 = d2; i1 = key; }
                else if (d2 < b2) b2 = d2;
            }
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    // merge the two lane halves (disjoint train rows)
    const float ob1 = __shfl_xor(b1, 32, 64), ob2 = __shfl_xor(b2, 32, 64);
    const int oi1 = __shfl_xor(i1, 32, 64);
    float m1, m2; int mi;
    if (ob1 < b1 || (ob1 == b1 && (unsigned)oi1 < (unsigned)i1)) { m1 = ob1; mi = oi1; m2 = fminf(b1, fminf(b2, ob2)); }
    else { m1 = b1; mi = i1; m2 = fminf(ob1, fminf(b2, ob2)); }
    if (qok && half == 0) {
        nn_idx[(size_t)b * N0 + q] = mi;
        nn_d2[((size_t)b * N0 + q) * 2] = m1;
        nn_d2[((size_t)b * N0 + q) * 2 + 1] = m2;
    }
}

// Lowe ratio test on the L2 distances (DMatch.distance = sqrt of FLANN's squared L2, f32; the Python
// comparison `m.distance < ratio * n.distance` is evaluated in binary64) + ordered compaction
__global__ void __launch_bounds__(256) desc_ratio_kernel(
    int N0, int N1, const int *__restrict__ n0v, const int *__restrict__ n1v, const int *__restrict__ nn_idx,
    const float *__restrict__ nn_d2, double ratio, const float *__restrict__ kp0, const float *__restrict__ kp1,
    float *__restrict__ pts0, float *__restrict__ pts1, int maxN, int *__restrict__ n_corr)
{
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = n0v[b], n = n1v[b];
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < m; start += 256) {
        const int i = start + tid;
        bool valid = false;
        int j = -1;
        if (i < m && n >= 2) {
            j = nn_idx[(size_t)b * N0 + i];
            const float d1 = sqrt_rn_f32(nn_d2[((size_t)b * N0 + i) * 2]);
            const float d2 = sqrt_rn_f32(nn_d2[((size_t)b * N0 + i) * 2 + 1]);
            valid = (double)d1 < ratio * (double)d2;
        }
        const unsigned long long bal = __ballot(valid);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wid; ++w) off += wave_cnt[w];
        if (valid) {
            const int o = off + wpre;
            if (o < maxN) {
                pts0[((size_t)b * maxN + o) * 2] = kp0[((size_t)b * N0 + i) * 2];
                pts0[((size_t)b * maxN + o) * 2 + 1] = kp0[((size_t)b * N0 + i) * 2 + 1];
                pts1[((size_t)b * maxN + o) * 2] = kp1[((size_t)b * N1 + j) * 2];
                pts1[((size_t)b * maxN + o) * 2 + 1] = kp1[((size_t)b * N1 + j) * 2 + 1];
            }
        }
        __syncthreads();
        if (tid == 0) base_s = off + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_corr[b] = min(base_s, maxN);
}

extern "C" {

int mfr_rootsift(const float *desc, int n_rows, float *out, float *norm2, void *stream)
{
    if (!desc || !out || !norm2 || n_rows < 0) return MFR_E_ARG;
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(rootsift_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, desc, n_rows, out, norm2);
    CHECK_LAUNCH();
    return 0;
}

int mfr_desc_ratio_match(const float *des0, const float *des1, const float *norm0, const float *norm1,
                         const float *kp0, const float *kp1, int B, int N0, int N1,
                         const int32_t *n0, const int32_t *n1, double ratio,
                         int32_t *nn_idx, float *nn_d2, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                         void *stream)
{
    if (!des0 || !des1 || !norm0 || !norm1 || !kp0 || !kp1 || !n0 || !n1 || !nn_idx || !nn_d2 || !pts0 || !pts1 || !n_corr ||
        B <= 0 || N0 <= 0 || N1 <= 0 || maxN <= 0)
        return MFR_E_ARG;
    const int nqb = (N0 + DM_QW * DM_WAVES - 1) / (DM_QW * DM_WAVES);
    hipLaunchKernelGGL(desc_2nn_kernel, dim3(nqb * B), dim3(256), 0, (hipStream_t)stream, des0, des1, norm0, norm1, B, N0, N1,
                       n0, n1, nn_idx, nn_d2);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(desc_ratio_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, N0, N1, n0, n1, nn_idx, nn_d2, ratio,
                       kp0, kp1, pts0, pts1, maxN, n_corr);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
