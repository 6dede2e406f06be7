// attention.hip -- SuperGlue multi-head softmax attention (self and cross) on gfx950, exact fp32.
//
// Reference call site: SuperGlue_matcher (etc/feature_matching_baselines/matchers.py:62-120) ->
// upstream AttentionalGNN / MultiHeadedAttention (un-vendored):
//     prob = softmax(q^T k / sqrt(64)) ; message = prob v        4 heads x 64, N <= 1024 keypoints,
// applied 18 x 2 times per image pair.  Upstream materialises the [4, N, N] score tensor
// (16.8 MB fp32 per application); here it never leaves registers (flash-style online softmax).
//
// Mapping to CDNA4: one wavefront owns 32 queries.  Both contractions run on the exact-fp32
// matrix cores (v_mfma_f32_32x32x2_f32: bit-identical to an fmaf chain, so no precision is traded
// against the fp32 reference):
//   S^T[key, q] = sum_d K[key,d] Q[q,d]   A = K tile (LDS, 16-B reads, row stride 68 floats:
//                                         conflict-free), B = Q^T held in 32 VGPRs for the whole loop
//   O^T[d, q]   = sum_key V[key,d] P[key,q]  A = V tile (LDS), B = P -- the probabilities are consumed
//                                         as the B operand IN the accumulator layout S^T was produced
//                                         in (row<->key pairing chosen to match), so P never moves.
// Softmax statistics are per query = per lane column: the row reduction is 16 in-register maxes /
// adds plus ONE cross-half exchange (lane ^ 32).  K/V tiles (32 keys) are prefetched into
// registers while the previous tile is being multiplied and double-buffered in LDS (one barrier
// per tile).  Keys >= n_tok[image] are masked; cross attention just reads the partner image's K/V
// (image b ^ 1), no copy.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AT_D 64
#define AT_KT 32            // keys per tile
#define AT_KS 68            // K tile row stride (floats): 272 B -> conflict-free ds_read_b128
#define AT_QW 32            // queries per wavefront
#define AT_WAVES 4

__global__ void __launch_bounds__(256, 2) sg_attention_kernel(
    const float *__restrict__ Q, const float *__restrict__ Kp, const float *__restrict__ Vp, int ld,
    int N, int heads, int B2, const int *__restrict__ n_tok, int cross, float scale_log2e, float *__restrict__ O, int ldo)
{
    __shared__ __attribute__((aligned(16))) float Ks[2][AT_KT][AT_KS];
    __shared__ __attribute__((aligned(16))) float Vs[2][AT_KT][AT_D];
    // 1-D grid, (image, head) fastest: workgroup L runs on XCD L % 8 (observed dispatch), so with
    // heads*B2 a multiple of 8 all query blocks of one (image, head) share an XCD and its K/V tiles are
    // fetched into that XCD's L2 once instead of once per query block (PMC: 5.6x the algorithmic reads
    // with the query block as the fastest index)
    const int nbh = heads * B2;
    const int bh = blockIdx.x % nbh, qb = blockIdx.x / nbh;
    const int b = bh / heads, h = bh - b * heads;
    const int bk = cross ? (b ^ 1) : b;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ql = lane & 31, half = lane >> 5;
    const int nq = n_tok[b], nk = n_tok[bk];
    const int q0 = qb * (AT_QW * AT_WAVES);
    const int q = q0 + wid * AT_QW + ql;
    if (q0 >= nq) {                                         // whole workgroup beyond this image's keypoints:
        if (q < N) {                                        // rows >= n_tok are defined to be zero
            float4 *op = (float4 *)(O + ((size_t)b * N + q) * ldo + h * AT_D + 32 * half);
#pragma unroll
            for (int g = 0; g < 8; ++g) op[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    // Q^T operand: lane (q, half) keeps Q[q][half*32 + s], s = 0..31, pre-scaled by log2(e)/sqrt(64)
    float qreg[32];
    {
        const bool ok = q < N;
        const float4 *qp = (const float4 *)(Q + ((size_t)b * N + (ok ? q : 0)) * ld + h * AT_D + half * 32);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float4 t = ok ? qp[g] : make_float4(0.f, 0.f, 0.f, 0.f);
            qreg[4 * g] = t.x * scale_log2e; qreg[4 * g + 1] = t.y * scale_log2e;
            qreg[4 * g + 2] = t.z * scale_log2e; qreg[4 * g + 3] = t.w * scale_log2e;
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging assignment: thread -> (row r and r+16, 4 floats at column c4*4)
    const int sr = tid >> 4, sc = (tid & 15) * 4;
    const float *kbase = Kp + (size_t)bk * N * ld + h * AT_D + sc;
    const float *vbase = Vp + (size_t)bk * N * ld + h * AT_D + sc;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    float4 kr0, kr1, vr0, vr1;
    auto gload = [&](int t) {
        const int k0 = t * AT_KT + sr, k1 = k0 + 16;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        kr0 = (k0 < nk) ? *(const float4 *)(kbase + (size_t)k0 * ld) : z;
        kr1 = (k1 < nk) ? *(const float4 *)(kbase + (size_t)k1 * ld) : z;
        vr0 = (k0 < nk) ? *(const float4 *)(vbase + (size_t)k0 * ld) : z;
        vr1 = (k1 < nk) ? *(const float4 *)(vbase + (size_t)k1 * ld) : z;
    };
    auto lstore = [&](int buf) {
        *(float4 *)&Ks[buf][sr][sc] = kr0; *(float4 *)&Ks[buf][sr + 16][sc] = kr1;
        *(float4 *)&Vs[buf][sr][sc] = vr0; *(float4 *)&Vs[buf][sr + 16][sc] = vr1;
    };
    if (ntiles > 0) { gload(0); lstore(0); }
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) gload(t + 1);                   // in flight during the MFMAs below

        // ---- S^T = K Q^T (32 keys x 32 queries, contraction 64 as 32 steps of 2)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float *krow = &Ks[buf][ql][half * 32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 a = *(const float4 *)(krow + 4 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * g], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * g + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * g + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * g + 3], s, 0, 0, 0);
        }
        // ---- online softmax over this tile's keys (rows of S^T); key = (r&3) + 8(r>>2) + 4 half
        const int kb = t * AT_KT + 4 * half;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2);
            if (key >= nk) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);               // finite: tile 0 always holds key 0
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); rs += s[r]; }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // ---- O^T += V^T P   (A = V[key(r,half)][d], B = p[r])
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 4 * half + (r & 3) + 8 * (r >> 2);
            const float a0 = Vs[buf][key][ql], a1 = Vs[buf][key][32 + ql];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
        }
        if (t + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }

    if (q < N) {
        const float inv = (l_run > 0.f && q < nq) ? 1.f / l_run : 0.f;
        float *op = O + ((size_t)b * N + q) * ldo + h * AT_D + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // rows (r&3) + 8(r>>2) + 4 half, r = 4g..4g+3 -> 4 consecutive d starting at 8g + 4 half
            *(float4 *)(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *(float4 *)(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

extern "C" {

// q,k,v: [B2, N, ld] fp32 (row = keypoint; channels of head h at [h*64, h*64+64) from the given base
// pointers, so a fused [.., 768] qkv buffer is passed as base, base+256, base+512 with ld = 768).
// out: [B2, N, ldo].  cross != 0: image b attends to image b^1 (the other image of its pair).
int mfr_sg_attention(const float *q, const float *k, const float *v, int ld, int B2, int N, int heads,
                     const int32_t *n_tok, int cross, float *out, int ldo, void *stream)
{
    if (!q || !k || !v || !n_tok || !out || B2 <= 0 || N <= 0 || heads <= 0 || (ld & 3) || (ldo & 3)) return MFR_E_ARG;
    if (cross && (B2 & 1)) return MFR_E_ARG;
    const float scale_log2e = 1.4426950408889634f / 8.0f;          // log2(e) / sqrt(64)
    const int nqb = (N + AT_QW * AT_WAVES - 1) / (AT_QW * AT_WAVES);
    dim3 grid(nqb * heads * B2);
    hipLaunchKernelGGL(sg_attention_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, ld, N, heads, B2, n_tok, cross,
                       scale_log2e, out, ldo);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
