// corr_warp.hip -- the relative-pose-regression aggregator ("correlation volume warping") on gfx950,
// forward and backward, without ever materialising the [N, N] correlation volume.
//
// Reference: CorrelationVolumeWarping.forward (lib/models/regression/aggregator.py:42-116) and
// CorrelationVolumeWarpingQKV.forward (aggregator.py:134-191):
//     cvolume = softmax(vol0^T vol1, dim=2)                 [B, N, N], N = H*W = 92*68 = 6256 on Map-free
//     vol1w   = vol1 cvolume^T                              warped features           [B, D, N]
//     pos     = grid cvolume^T                              expected (u, v) position  [B, 2, N]
//     max     = max_j cvolume[:, i, j]                      matching confidence       [B, 1, N]
// The reference writes cvolume (1.57 GB fp32 at batch 10) and reads it back 3-4 times forward and again in
// autograd's backward.  Here every query block streams the keys once per pass (flash style, online
// softmax); the volume lives in MFMA accumulators only.  max_j softmax = 1 / sum_j exp(s_j - max s), so
// the confidence channel is free.  All contractions run on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32); the whole aggregator is ~0.2 TFLOP per training step, the point is the 20+ GB
// of volume traffic that disappears.
//
// Layout: tensors are channels-first [B, D, N] exactly as the encoder produces them (N contiguous), tiles
// of 32 tokens are staged as [channel][token] rows of 33 floats: conflict-free both when a lane reads
// "its token" (A operand of S = K^T Q) and when it reads "its channel" (A operand of O = V P).
//
// Backward (same tiling, two kernels so that every output has ONE owner and no atomics):
//     P_ij  = exp2(s_ij - m_i) / l_i      (m, l saved by the forward; s recomputed bit-identically)
//     dP_ij = dO_i . V_j + dpos_i . grid_j + dmax_i [s_ij == m_i]
//     dS_ij = P_ij (dP_ij - delta_i),     delta_i = dO_i . O_i + dpos_i . pos_i + dmax_i max_i
//     cw_bwd_q_kernel  (owner: 32 queries / wave, loops over keys):     dQ_i = sum_j dS_ij K_j
//     cw_bwd_kv_kernel (owner: 32 keys / wave, loops over queries):     dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CW_T 32             // tokens per tile
#define CW_LD 33            // tile row stride (floats)
#define CW_DV 32            // value channels (ENCODER.NUM_OUT_LAYERS of every shipped config)
#define CW_LOG2E 1.4426950408889634f

// accumulator row r of lane half h <-> tile row (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ int cw_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---------------------------------------------------------------------------------------------------------
// forward
template <int DQ>
__global__ void __launch_bounds__(256, 2) cw_fwd_kernel(
    const float *__restrict__ Q, const float *__restrict__ K, const float *__restrict__ V, const float *__restrict__ G,
    int B, int N, float *__restrict__ Wout, float *__restrict__ Pout, float *__restrict__ MS, float *__restrict__ RM,
    float *__restrict__ RS)
{
    __shared__ float Ks[2][DQ][CW_LD];
    __shared__ float Vs[2][CW_DV][CW_LD];
    __shared__ __attribute__((aligned(16))) float Gs[2][2][CW_T];
    const int b = blockIdx.x % B, qb = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, ql = lane & 31, half = lane >> 5;
    const int q = qb * 128 + wid * 32 + ql;
    const bool qok = q < N;
    const float *Qb = Q + (size_t)b * DQ * N, *Kb = K + (size_t)b * DQ * N, *Vb = V + (size_t)b * CW_DV * N;

    float qreg[DQ / 2];
#pragma unroll
    for (int s = 0; s < DQ / 2; ++s) qreg[s] = qok ? Qb[(size_t)(2 * s + half) * N + q] * CW_LOG2E : 0.f;

    // staging: thread -> token column sc, rows sr + 8 i
    const int sc = tid & 31, sr = tid >> 5;
    float kr[DQ / 8], vr[CW_DV / 8], gr = 0.f;
    auto gload = [&](int t) {
        const int key = t * CW_T + sc;
        const bool ok = key < N;
#pragma unroll
        for (int i = 0; i < DQ / 8; ++i) kr[i] = ok ? Kb[(size_t)(sr + 8 * i) * N + key] : 0.f;
#pragma unroll
        for (int i = 0; i < CW_DV / 8; ++i) vr[i] = ok ? Vb[(size_t)(sr + 8 * i) * N + key] : 0.f;
        if (G && tid < 64) gr = ok ? G[(size_t)(tid >> 5) * N + key] : 0.f;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DQ / 8; ++i) Ks[buf][sr + 8 * i][sc] = kr[i];
#pragma unroll
        for (int i = 0; i < CW_DV / 8; ++i) Vs[buf][sr + 8 * i][sc] = vr[i];
        if (tid < 64) Gs[buf][tid >> 5][sc] = gr;
    };
    const int ntiles = (N + CW_T - 1) / CW_T;
    gload(0); lstore(0);
    __syncthreads();

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f, pu = 0.f, pv = 0.f;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        // S^T[key, q] = sum_d K[d, key] Q[d, q]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < DQ / 2; ++st)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][2 * st + half][ql], qreg[st], s, 0, 0, 0);
        const int kb = t * CW_T;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (kb + cw_row(r, half) >= N) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs = 0.f, tu = 0.f, tv = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 gu = *(const float4 *)&Gs[buf][0][8 * g + 4 * half];
            const float4 gv = *(const float4 *)&Gs[buf][1][8 * g + 4 * half];
            const float u4[4] = {gu.x, gu.y, gu.z, gu.w}, v4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = __builtin_amdgcn_exp2f(s[4 * g + j] - m_new);
                s[4 * g + j] = p; rs += p;
                tu = fmaf(p, u4[j], tu); tv = fmaf(p, v4[j], tv);
            }
        }
        l_run = l_run * alpha + rs; pu = pu * alpha + tu; pv = pv * alpha + tv;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        // O^T[c, q] += sum_key V[c, key] P[key, q]
#pragma unroll
        for (int r = 0; r < 16; ++r)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][ql][cw_row(r, half)], s[r], o, 0, 0, 0);
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    l_run += __shfl_xor(l_run, 32, 64);
    pu += __shfl_xor(pu, 32, 64);
    pv += __shfl_xor(pv, 32, 64);
    if (qok) {
        const float inv = 1.f / l_run;
        float *wp = Wout + (size_t)b * CW_DV * N + q;
#pragma unroll
        for (int r = 0; r < 16; ++r) wp[(size_t)cw_row(r, half) * N] = o[r] * inv;
        if (half == 0) {
            if (Pout) { Pout[((size_t)b * 2) * N + q] = pu * inv; Pout[((size_t)b * 2 + 1) * N + q] = pv * inv; }
            MS[(size_t)b * N + q] = inv; RM[(size_t)b * N + q] = m_run; RS[(size_t)b * N + q] = l_run;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, query owner: dQ
template <int DQ>
__global__ void __launch_bounds__(256, 2) cw_bwd_q_kernel(
    const float *__restrict__ Q, const float *__restrict__ K, const float *__restrict__ V, const float *__restrict__ G,
    int B, int N, const float *__restrict__ dW, const float *__restrict__ dP, const float *__restrict__ dM,
    const float *__restrict__ DL, const float *__restrict__ RM, const float *__restrict__ RS, float *__restrict__ dQ)
{
    __shared__ float Ks[2][DQ][CW_LD];
    __shared__ float Vs[2][CW_DV][CW_LD];
    __shared__ __attribute__((aligned(16))) float Gs[2][2][CW_T];
    const int b = blockIdx.x % B, qb = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, ql = lane & 31, half = lane >> 5;
    const int q = qb * 128 + wid * 32 + ql;
    const bool qok = q < N;
    const float *Qb = Q + (size_t)b * DQ * N, *Kb = K + (size_t)b * DQ * N, *Vb = V + (size_t)b * CW_DV * N;
    const float *dWb = dW + (size_t)b * CW_DV * N;

    float qreg[DQ / 2], doreg[CW_DV / 2];
#pragma unroll
    for (int s = 0; s < DQ / 2; ++s) qreg[s] = qok ? Qb[(size_t)(2 * s + half) * N + q] * CW_LOG2E : 0.f;
#pragma unroll
    for (int s = 0; s < CW_DV / 2; ++s) doreg[s] = qok ? dWb[(size_t)(2 * s + half) * N + q] : 0.f;
    const size_t bq = (size_t)b * N + (qok ? q : 0);
    const float m_q = RM[bq], invl = qok ? 1.f / RS[bq] : 0.f, delta = DL[bq];
    const float gms = dM ? dM[bq] : 0.f;
    const float dpu = dP ? dP[((size_t)b * 2) * N + (qok ? q : 0)] : 0.f, dpv = dP ? dP[((size_t)b * 2 + 1) * N + (qok ? q : 0)] : 0.f;

    const int sc = tid & 31, sr = tid >> 5;
    float kr[DQ / 8], vr[CW_DV / 8], gr = 0.f;
    auto gload = [&](int t) {
        const int key = t * CW_T + sc;
        const bool ok = key < N;
#pragma unroll
        for (int i = 0; i < DQ / 8; ++i) kr[i] = ok ? Kb[(size_t)(sr + 8 * i) * N + key] : 0.f;
#pragma unroll
        for (int i = 0; i < CW_DV / 8; ++i) vr[i] = ok ? Vb[(size_t)(sr + 8 * i) * N + key] : 0.f;
        if (G && tid < 64) gr = ok ? G[(size_t)(tid >> 5) * N + key] : 0.f;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DQ / 8; ++i) Ks[buf][sr + 8 * i][sc] = kr[i];
#pragma unroll
        for (int i = 0; i < CW_DV / 8; ++i) Vs[buf][sr + 8 * i][sc] = vr[i];
        if (tid < 64) Gs[buf][tid >> 5][sc] = gr;
    };
    const int ntiles = (N + CW_T - 1) / CW_T;
    gload(0); lstore(0);
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int dql = ql & (DQ - 1);

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < DQ / 2; ++st)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][2 * st + half][ql], qreg[st], s, 0, 0, 0);
        // dP^T[key, q] = sum_c V[c, key] dO[c, q]
#pragma unroll
        for (int st = 0; st < CW_DV / 2; ++st)
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][2 * st + half][ql], doreg[st], dp, 0, 0, 0);
        const int kb = t * CW_T;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 gu = *(const float4 *)&Gs[buf][0][8 * g + 4 * half];
            const float4 gv = *(const float4 *)&Gs[buf][1][8 * g + 4 * half];
            const float u4[4] = {gu.x, gu.y, gu.z, gu.w}, v4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                const bool kok = kb + cw_row(r, half) < N;
                const float p = kok ? __builtin_amdgcn_exp2f(s[r] - m_q) * invl : 0.f;
                float e = dp[r] - delta;
                e = fmaf(dpu, u4[j], e); e = fmaf(dpv, v4[j], e);
                if (s[r] == m_q) e += gms;
                s[r] = p * e;                                   // dS^T[key, q]
            }
        }
        // dQ^T[d, q] += sum_key K[d, key] dS^T[key, q]
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][dql][cw_row(r, half)], s[r], acc, 0, 0, 0);
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    if (qok) {
        float *op = dQ + (size_t)b * DQ * N + q;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = cw_row(r, half);
            if (d < DQ) op[(size_t)d * N] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward, key owner: dK and dV
template <int DQ>
__global__ void __launch_bounds__(256, 2) cw_bwd_kv_kernel(
    const float *__restrict__ Q, const float *__restrict__ K, const float *__restrict__ V, const float *__restrict__ G,
    int B, int N, const float *__restrict__ dW, const float *__restrict__ dP, const float *__restrict__ dM,
    const float *__restrict__ DL, const float *__restrict__ RM, const float *__restrict__ RS, float *__restrict__ dK,
    float *__restrict__ dV)
{
    __shared__ float Qs[2][DQ][CW_LD];
    __shared__ float Ds[2][CW_DV][CW_LD];
    __shared__ __attribute__((aligned(16))) float Rs[2][CW_T][8];          // per query: m, 1/l, delta, dmax, dpos_u, dpos_v
    const int b = blockIdx.x % B, kblk = blockIdx.x / B;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, ql = lane & 31, half = lane >> 5;
    const int key = kblk * 128 + wid * 32 + ql;
    const bool kok = key < N;
    const float *Qb = Q + (size_t)b * DQ * N, *Kb = K + (size_t)b * DQ * N, *Vb = V + (size_t)b * CW_DV * N;
    const float *dWb = dW + (size_t)b * CW_DV * N;

    float kreg[DQ / 2], vreg[CW_DV / 2];
#pragma unroll
    for (int s = 0; s < DQ / 2; ++s) kreg[s] = kok ? Kb[(size_t)(2 * s + half) * N + key] : 0.f;
#pragma unroll
    for (int s = 0; s < CW_DV / 2; ++s) vreg[s] = kok ? Vb[(size_t)(2 * s + half) * N + key] : 0.f;
    const float gu = (G && kok) ? G[key] : 0.f, gv = (G && kok) ? G[(size_t)N + key] : 0.f;

    const int sc = tid & 31, sr = tid >> 5;
    float qr[DQ / 8], dr[CW_DV / 8], rr = 0.f;
    auto gload = [&](int t) {
        const int q = t * CW_T + sc;
        const bool ok = q < N;
#pragma unroll
        for (int i = 0; i < DQ / 8; ++i) qr[i] = ok ? Qb[(size_t)(sr + 8 * i) * N + q] : 0.f;
#pragma unroll
        for (int i = 0; i < CW_DV / 8; ++i) dr[i] = ok ? dWb[(size_t)(sr + 8 * i) * N + q] : 0.f;
        const size_t bq = (size_t)b * N + (ok ? q : 0);
        switch (sr) {                                           // one scalar kind per staging row
        case 0: rr = ok ? RM[bq] : 0.f; break;
        case 1: rr = ok ? 1.f / RS[bq] : 0.f; break;
        case 2: rr = ok ? DL[bq] : 0.f; break;
        case 3: rr = (ok && dM) ? dM[bq] : 0.f; break;
        case 4: rr = (ok && dP) ? dP[((size_t)b * 2) * N + q] : 0.f; break;
        case 5: rr = (ok && dP) ? dP[((size_t)b * 2 + 1) * N + q] : 0.f; break;
        default: rr = 0.f;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DQ / 8; ++i) Qs[buf][sr + 8 * i][sc] = qr[i];
#pragma unroll
        for (int i = 0; i < CW_DV / 8; ++i) Ds[buf][sr + 8 * i][sc] = dr[i];
        Rs[buf][sc][sr] = rr;
    };
    const int ntiles = (N + CW_T - 1) / CW_T;
    gload(0); lstore(0);
    __syncthreads();

    f32x16 ak, av;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ak[r] = 0.f; av[r] = 0.f; }
    const int dql = ql & (DQ - 1);

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S[q, key]: same products, same order over d as the forward -> bit-identical scores
#pragma unroll
        for (int st = 0; st < DQ / 2; ++st)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[buf][2 * st + half][ql] * CW_LOG2E, kreg[st], s, 0, 0, 0);
        // dP[q, key] = sum_c dO[c, q] V[c, key]
#pragma unroll
        for (int st = 0; st < CW_DV / 2; ++st)
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[buf][2 * st + half][ql], vreg[st], dp, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = cw_row(r, half);
            const float4 r0 = *(const float4 *)&Rs[buf][qq][0];     // m, 1/l, delta, dmax
            const float2 r1 = *(const float2 *)&Rs[buf][qq][4];     // dpos_u, dpos_v
            const float p = __builtin_amdgcn_exp2f(s[r] - r0.x) * r0.y;
            float e = dp[r] - r0.z;
            e = fmaf(r1.x, gu, e); e = fmaf(r1.y, gv, e);
            if (s[r] == r0.x) e += r0.w;
            dp[r] = p;                                          // P[q, key]
            s[r] = p * e;                                       // dS[q, key]
        }
        // dK^T[d, key] += sum_q Q[d, q] dS[q, key];   dV^T[c, key] += sum_q dO[c, q] P[q, key]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = cw_row(r, half);
            ak = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[buf][dql][qq], s[r], ak, 0, 0, 0);
            av = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[buf][ql][qq], dp[r], av, 0, 0, 0);
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    if (kok) {
        float *kp = dK + (size_t)b * DQ * N + key, *vp = dV + (size_t)b * CW_DV * N + key;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = cw_row(r, half);
            if (d < DQ) kp[(size_t)d * N] = ak[r];
            vp[(size_t)d * N] = av[r];
        }
    }
}

extern "C" {

// q, k: [B, Dq, N]  v: [B, 32, N]  grid: [2, N] or NULL (POSITION_ENCODER off).  Outputs: warped [B, 32, N],
// pos [B, 2, N] (NULL iff grid NULL), max_score [B, N], and the softmax statistics the backward needs:
// row_max [B, N] (of log2(e) * score) and row_sum [B, N].
int mfr_corr_warp_fwd(const float *q, const float *k, const float *v, const float *grid, int B, int Dq, int N,
                      float *warped, float *pos, float *max_score, float *row_max, float *row_sum, void *stream)
{
    if (!q || !k || !v || !warped || !max_score || !row_max || !row_sum || B <= 0 || N <= 0) return MFR_E_ARG;
    if ((Dq != 16 && Dq != 32) || ((grid == nullptr) != (pos == nullptr))) return MFR_E_ARG;
    const dim3 g(B * ((N + 127) / 128));
    if (Dq == 32)
        hipLaunchKernelGGL(cw_fwd_kernel<32>, g, dim3(256), 0, (hipStream_t)stream, q, k, v, grid, B, N, warped, pos, max_score, row_max, row_sum);
    else
        hipLaunchKernelGGL(cw_fwd_kernel<16>, g, dim3(256), 0, (hipStream_t)stream, q, k, v, grid, B, N, warped, pos, max_score, row_max, row_sum);
    CHECK_LAUNCH();
    return 0;
}

// d_warped [B, 32, N], d_pos [B, 2, N] or NULL, d_max [B, N] or NULL: incoming gradients.  delta [B, N] =
// sum_c d_warped*warped + sum d_pos*pos + d_max*max_score (the softmax-Jacobian row term, a pointwise
// reduction the caller already has the operands for).  Outputs dq, dk [B, Dq, N], dv [B, 32, N].
int mfr_corr_warp_bwd(const float *q, const float *k, const float *v, const float *grid, int B, int Dq, int N,
                      const float *d_warped, const float *d_pos, const float *d_max, const float *delta,
                      const float *row_max, const float *row_sum, float *dq, float *dk, float *dv, void *stream)
{
    if (!q || !k || !v || !d_warped || !delta || !row_max || !row_sum || !dq || !dk || !dv || B <= 0 || N <= 0) return MFR_E_ARG;
    if ((Dq != 16 && Dq != 32) || (d_pos && !grid)) return MFR_E_ARG;
    const dim3 g(B * ((N + 127) / 128));
    if (Dq == 32) {
        hipLaunchKernelGGL(cw_bwd_q_kernel<32>, g, dim3(256), 0, (hipStream_t)stream, q, k, v, grid, B, N, d_warped, d_pos, d_max, delta, row_max, row_sum, dq);
        hipLaunchKernelGGL(cw_bwd_kv_kernel<32>, g, dim3(256), 0, (hipStream_t)stream, q, k, v, grid, B, N, d_warped, d_pos, d_max, delta, row_max, row_sum, dk, dv);
    } else {
        hipLaunchKernelGGL(cw_bwd_q_kernel<16>, g, dim3(256), 0, (hipStream_t)stream, q, k, v, grid, B, N, d_warped, d_pos, d_max, delta, row_max, row_sum, dq);
        hipLaunchKernelGGL(cw_bwd_kv_kernel<16>, g, dim3(256), 0, (hipStream_t)stream, q, k, v, grid, B, N, d_warped, d_pos, d_max, delta, row_max, row_sum, dk, dv);
    }
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
