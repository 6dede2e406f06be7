// attention.hip -- SuperGlue multi-head softmax attention (self and cross) on gfx950: the exact-fp32 kernel (rounds 1-2, kept for A/B and parity),
// and the operand-splitting kernels on the 16-bit matrix cores at fp32 accuracy (bf16x3: rounds 3-4; f16x2: round 5, the default).
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
#include "split_f16.h"
#include "guard.h"

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


// ------------------------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix cores at fp32 accuracy ("bf16x3"): every fp32 operand x is split EXACTLY into
// three bf16 terms x = h + m + l (truncation: 8 + 8 + 8 significand bits), and a product a.b is evaluated as the six
// partial products hh + hm + mh + hl + lh + mm (each exact in fp32) accumulated in fp32 by v_mfma_f32_32x32x16_bf16 --
// the dropped terms (ml, lm, ll) are below 2^-26 |a||b|.  Measured against an fp64 product (tools/ubench/bf16x3_probe.hip,
// profiles/r03_bf16x3_probe.jsonl; K = 64 .. 2304): rms / max error 2.4e-8 / 2.3e-7 of sum|a||b|, vs 2.8e-8 / 2.6e-7 for the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) on the same data -- the same error class, at 16/6 = 2.7x the matrix rate.
// Structure as above (one wavefront = 32 queries, 32-key tiles, online softmax in the accumulator layout, P never moves):
//   S^T = K Q^T   A = K tile split in LDS (row stride 72 bf16: conflict-free 16-B reads), B = Q^T split once into 48 VGPRs
//   O^T = V^T P   A = V^T tile split in LDS, keys stored in the order the S^T accumulator hands them out (position
//                 16s + 8h + 4g + j for key 16s + 8g + 4h + j), so a lane's eight contraction slots are one 16-B read;
//                 B = P split in registers (16 values per lane per tile)
// The K / V tiles are split once per workgroup by the staging threads (V is fetched d-per-lane so that its transpose
// is two 8-byte LDS stores per term).
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

#define AB_KS 72            // K tile row stride (bf16): 144 B
#define AB_VS 40            // V^T tile row stride (bf16): 80 B

__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    // upper 16 bits of each term = its bf16 pattern; h + m + l == x exactly
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
// pack the bf16 (upper) halves of two fp32 bit patterns: lo -> bits 15:0, hi -> bits 31:16
__device__ __forceinline__ unsigned pack_hi16(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// two values at a time: the subtractions are packed fp32 instructions (v_pk_add_f32 with a negated operand, one issue slot for two lanes of data)
typedef float at_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(at_f2 x, unsigned &ph, unsigned &pm, unsigned &pl)
{
    const unsigned x0 = __float_as_uint(x.x), x1 = __float_as_uint(x.y);
    at_f2 h; h.x = __uint_as_float(x0 & 0xffff0000u); h.y = __uint_as_float(x1 & 0xffff0000u);
    const at_f2 r = x - h;
    const unsigned r0 = __float_as_uint(r.x), r1 = __float_as_uint(r.y);
    at_f2 m; m.x = __uint_as_float(r0 & 0xffff0000u); m.y = __uint_as_float(r1 & 0xffff0000u);
    const at_f2 l = r - m;
    ph = pack_hi16(x0, x1); pm = pack_hi16(r0, r1); pl = pack_hi16(__float_as_uint(l.x), __float_as_uint(l.y));
}
template <int N8>
__device__ __forceinline__ void split_pack(const float (&x)[N8], unsigned (&ph)[N8 / 2], unsigned (&pm)[N8 / 2], unsigned (&pl)[N8 / 2])
{
#pragma unroll
    for (int j = 0; j < N8 / 2; ++j) {
        at_f2 v; v.x = x[2 * j]; v.y = x[2 * j + 1];
        split_pair(v, ph[j], pm[j], pl[j]);
    }
}

union Frag8 { bf16x8 v; unsigned u[4]; uint4 q; };

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---- round 4 (variant 2 since round 5): eight wavefronts (256 queries) per workgroup, software-pipelined -----------------------------------------
// K / V rows are read through buffer descriptors that end at row nk: the row offset of a tile is a SCALAR (no address VALU), rows beyond nk
// read as zero in hardware (no clamp, no select); a K / V tile is split once per 256 queries.  (Round 3's 128-query kernel and the
// non-pipelined eight-wavefront kernel computed the same bits and left the library in round 5: profiles/r04_bench_attention.json.)
#define AB_WAVES 8
// Without pipelining a wavefront's tile is [24 MFMAs: S^T] -> [softmax: ~120 VALU slots, no MFMA] -> [24 MFMAs: O^T, with the split of P];
// the barrier per tile keeps all wavefronts of the CU in the same phase, so the matrix core idles through every softmax (measured: a tile
// costs a SIMD the SUM of its two wavefronts' MFMA and VALU time, 5400 cycles for 3072 of MFMA).  Here the score product runs one tile ahead:
// iteration t issues S^T(t+1) = K(t+1) Q^T in four steps of six MFMAs, and between them the softmax of tile t (whose scores were finished an
// iteration ago) -- independent instruction streams that one wavefront overlaps by itself; then O^T += V(t)^T P(t).  K is therefore staged one
// tile further ahead than V (K(t+2) and V(t+1) are written during iteration t; two LDS stages each, as before).  Per query the same
// arithmetic in the same order as the two kernels above.
__global__ void __launch_bounds__(512, 1) sg_attention_bf16x3_p_kernel(
    const float *__restrict__ Q, const float *__restrict__ Kp, const float *__restrict__ Vp, int ld,
    int N, int heads, int B2, const int *__restrict__ n_tok, int cross, float scale_log2e, float *__restrict__ O, int ldo)
{
    __shared__ __attribute__((aligned(16))) unsigned short Ks[2][3][AT_KT][AB_KS];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[2][3][AT_D][AB_VS];
    const int nbh = heads * B2;
    const int bh = blockIdx.x % nbh, qb = blockIdx.x / nbh;
    const int b = bh / heads, h = bh - b * heads;
    const int bk = cross ? (b ^ 1) : b;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int nq = n_tok[b], nk = n_tok[bk];
    const int q0 = qb * (AT_QW * AB_WAVES);
    const int q = q0 + wid * AT_QW + ql;
    if (q0 >= nq) {
        if (q < N) {
            float4 *op = (float4 *)(O + ((size_t)b * N + q) * ldo + h * AT_D + 32 * half);
#pragma unroll
            for (int g = 0; g < 8; ++g) op[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    Frag8 qf[4][3];
    {
        const bool ok = q < N;
        const float *qp = Q + ((size_t)b * N + (ok ? q : 0)) * ld + h * AT_D + half * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float x[8];
            float4 t0 = *(const float4 *)(qp + 16 * s), t1 = *(const float4 *)(qp + 16 * s + 4);
            if (!ok) { t0 = make_float4(0.f, 0.f, 0.f, 0.f); t1 = t0; }
            x[0] = t0.x * scale_log2e; x[1] = t0.y * scale_log2e; x[2] = t0.z * scale_log2e; x[3] = t0.w * scale_log2e;
            x[4] = t1.x * scale_log2e; x[5] = t1.y * scale_log2e; x[6] = t1.z * scale_log2e; x[7] = t1.w * scale_log2e;
            split_pack<8>(x, qf[s][0].u, qf[s][1].u, qf[s][2].u);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int sr = tid >> 4, sc4 = (tid & 15) * 4;
    const unsigned rowb = (unsigned)ld * 4u;
    const unsigned span = nk > 0 ? (unsigned)(nk - 1) * rowb + AT_D * 4u : 0u;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(Kp + (size_t)bk * N * ld + h * AT_D), 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(Vp + (size_t)bk * N * ld + h * AT_D), 0, (int)span, 0x00020000);
    const unsigned koff = (unsigned)sr * rowb + 4u * (unsigned)sc4, voff = 4u * (unsigned)lane;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    float4 kr;
    float v0, v1, v2, v3;
    auto gload_k = [&](int t) { kr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, (unsigned)(t * AT_KT) * rowb, 0)); };
    auto gload_v = [&](int t) {
        const unsigned sv = (unsigned)(t * AT_KT + 4 * wid) * rowb;
        v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv, 0));
        v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv + rowb, 0));
        v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv + 2u * rowb, 0));
        v3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv + 3u * rowb, 0));
    };
    auto lstore_k = [&](int buf) {
        unsigned ph[2], pm[2], pl[2];
        const float ka[4] = { kr.x, kr.y, kr.z, kr.w };
        split_pack<4>(ka, ph, pm, pl);
        *(uint2 *)&Ks[buf][0][sr][sc4] = make_uint2(ph[0], ph[1]); *(uint2 *)&Ks[buf][1][sr][sc4] = make_uint2(pm[0], pm[1]);
        *(uint2 *)&Ks[buf][2][sr][sc4] = make_uint2(pl[0], pl[1]);
    };
    auto lstore_v = [&](int buf) {
        unsigned ph[2], pm[2], pl[2];
        const float vr[4] = { v0, v1, v2, v3 };
        split_pack<4>(vr, ph, pm, pl);
        const int p0 = 16 * (wid >> 2) + 8 * (wid & 1) + 4 * ((wid >> 1) & 1);
        *(uint2 *)&Vt[buf][0][lane][p0] = make_uint2(ph[0], ph[1]); *(uint2 *)&Vt[buf][1][lane][p0] = make_uint2(pm[0], pm[1]);
        *(uint2 *)&Vt[buf][2][lane][p0] = make_uint2(pl[0], pl[1]);
    };
    const unsigned short *kq0 = &Ks[0][0][ql][8 * half], *vq0 = &Vt[0][0][ql][8 * half];
    // six MFMAs of step st of S^T = K Q^T from the K stage at kq
#define AP_QK_STEP(kq, st, s, s2) do { \
        Frag8 kh, km, kl; \
        kh.q = *(const uint4 *)((kq) + 16 * (st)); km.q = *(const uint4 *)((kq) + AT_KT * AB_KS + 16 * (st)); kl.q = *(const uint4 *)((kq) + 2 * AT_KT * AB_KS + 16 * (st)); \
        s2 = MFMA_BF16(km.v, qf[st][1].v, s2); s = MFMA_BF16(kh.v, qf[st][2].v, s); s2 = MFMA_BF16(kl.v, qf[st][0].v, s2); \
        s = MFMA_BF16(kh.v, qf[st][1].v, s); s2 = MFMA_BF16(km.v, qf[st][0].v, s2); s = MFMA_BF16(kh.v, qf[st][0].v, s); } while (0)

    f32x16 sc;                                              // the scores of the tile whose softmax is due
    gload_k(0); gload_v(0); lstore_k(0); lstore_v(0);       // (tiles beyond nk: zeros)
    gload_k(1); lstore_k(1);
    gload_k(2); gload_v(1);                                 // staged during iteration 0
    __syncthreads();
    {
        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
        if (ntiles > 0) { AP_QK_STEP(kq0, 0, s, s2); AP_QK_STEP(kq0, 1, s, s2); AP_QK_STEP(kq0, 2, s, s2); AP_QK_STEP(kq0, 3, s, s2); }
        sc = s + s2;
    }
    __syncthreads();                                        // K stage 0 is rewritten during iteration 0

    // fragments of S^T's step st (K stage at kq) / the six MFMAs on them; one MFMA then up to nv VALU, six times (sched_group_barrier)
#define AP_KLOAD(f, kq, st) do { f[0].q = *(const uint4 *)((kq) + 16 * (st)); f[1].q = *(const uint4 *)((kq) + AT_KT * AB_KS + 16 * (st)); \
        f[2].q = *(const uint4 *)((kq) + 2 * AT_KT * AB_KS + 16 * (st)); } while (0)
#define AP_QK6(f, st, s, s2) do { \
        s2 = MFMA_BF16(f[1].v, qf[st][1].v, s2); s = MFMA_BF16(f[0].v, qf[st][2].v, s); s2 = MFMA_BF16(f[2].v, qf[st][0].v, s2); \
        s = MFMA_BF16(f[0].v, qf[st][1].v, s); s2 = MFMA_BF16(f[1].v, qf[st][0].v, s2); s = MFMA_BF16(f[0].v, qf[st][0].v, s); } while (0)
#define AP_WEAVE(nm, nv) do { _Pragma("unroll") for (int w_ = 0; w_ < (nm); ++w_) { \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, (nv), 0); } } while (0)
#define AP_VLOAD(f, vq, st) do { \
        f[0].q = *(const uint4 *)((vq) + 16 * (st)); f[1].q = *(const uint4 *)((vq) + 32 * AB_VS + 16 * (st)); \
        f[2].q = *(const uint4 *)((vq) + AT_D * AB_VS + 16 * (st)); f[3].q = *(const uint4 *)((vq) + AT_D * AB_VS + 32 * AB_VS + 16 * (st)); \
        f[4].q = *(const uint4 *)((vq) + 2 * AT_D * AB_VS + 16 * (st)); f[5].q = *(const uint4 *)((vq) + 2 * AT_D * AB_VS + 32 * AB_VS + 16 * (st)); } while (0)
    // f: 0 / 1 = h term of channels 0-31 / 32-63, 2 / 3 = m, 4 / 5 = l
#define AP_PV12(f, ph, pm, pl) do { \
        o0 = MFMA_BF16(f[2].v, pm.v, o0); o1 = MFMA_BF16(f[3].v, pm.v, o1); o0 = MFMA_BF16(f[0].v, pl.v, o0); o1 = MFMA_BF16(f[1].v, pl.v, o1); \
        o0 = MFMA_BF16(f[4].v, ph.v, o0); o1 = MFMA_BF16(f[5].v, ph.v, o1); o0 = MFMA_BF16(f[0].v, pm.v, o0); o1 = MFMA_BF16(f[1].v, pm.v, o1); \
        o0 = MFMA_BF16(f[2].v, ph.v, o0); o1 = MFMA_BF16(f[3].v, ph.v, o1); o0 = MFMA_BF16(f[0].v, ph.v, o0); o1 = MFMA_BF16(f[1].v, ph.v, o1); } while (0)
    // The last iteration multiplies a K stage of zeros (tile ntiles does not exist) and stages tiles that are never used: one basic block per
    // chunk, so that the MFMAs and the VALU work can be woven together, is worth more than the 24 MFMAs it wastes.
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const unsigned short *kq = kq0 + (buf ^ 1) * (3 * AT_KT * AB_KS);
        const unsigned short *vq = vq0 + buf * (3 * AT_D * AB_VS);
        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
        Frag8 ka[3], kb2[3];
        AP_KLOAD(ka, kq, 0);
        AP_KLOAD(kb2, kq, 1);
        const int kb = t * AT_KT + 4 * half;
        if ((t + 1) * AT_KT > nk) {                         // only the last tile can hold keys >= nk (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb + (r & 3) + 8 * (r >> 2) >= nk) sc[r] = -INFINITY;
        }
        // ---- S^T(t+1), step 0  ||  softmax(t): row maximum
        AP_QK6(ka, 0, s, s2);
        AP_KLOAD(ka, kq, 2);
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
        mx = fmaxf(mx, sc[15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        AP_WEAVE(6, 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 1  ||  exponentials of keys 0 .. 7
        AP_QK6(kb2, 1, s, s2);
        AP_KLOAD(kb2, kq, 3);
        float p[16];
        at_f2 rs2; rs2.x = 0.f; rs2.y = 0.f;
        at_f2 mm; mm.x = m_new; mm.y = m_new;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            at_f2 d; d.x = sc[r]; d.y = sc[r + 1];
            d = d - mm;
            at_f2 e; e.x = __builtin_amdgcn_exp2f(d.x); e.y = __builtin_amdgcn_exp2f(d.y);
            rs2 += e;
            p[r] = e.x; p[r + 1] = e.y;
        }
        AP_WEAVE(6, 3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 2  ||  exponentials of keys 8 .. 15
        AP_QK6(ka, 2, s, s2);
#pragma unroll
        for (int r = 8; r < 16; r += 2) {
            at_f2 d; d.x = sc[r]; d.y = sc[r + 1];
            d = d - mm;
            at_f2 e; e.x = __builtin_amdgcn_exp2f(d.x); e.y = __builtin_amdgcn_exp2f(d.y);
            rs2 += e;
            p[r] = e.x; p[r + 1] = e.y;
        }
        float rs = rs2.x + rs2.y;
        rs += __shfl_xor(rs, 32, 64);
        AP_WEAVE(6, 3);
        __builtin_amdgcn_sched_barrier(0);
        if (__ballot(m_new != m_run) != 0ull) {             // the running maximum moved for some query of this wavefront: rescale
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        l_run += rs;
        m_run = m_new;
        // ---- step 3  ||  split of P (first 16 keys), staging of K(t+2) / V(t+1)
        AP_QK6(kb2, 3, s, s2);
        Frag8 va[6];
        AP_VLOAD(va, vq, 0);
        Frag8 ph0, pm0, pl0;
        {
            float pp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pp[j] = p[j];
            split_pack<8>(pp, ph0.u, pm0.u, pl0.u);
        }
        lstore_k(buf); lstore_v(buf ^ 1);
        gload_k(t + 3); gload_v(t + 2);
        AP_WEAVE(6, 14);
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T P: keys 0 .. 15  ||  split of P (last 16 keys); then keys 16 .. 31
        AP_PV12(va, ph0, pm0, pl0);
        Frag8 vb[6];
        AP_VLOAD(vb, vq, 1);
        Frag8 ph1, pm1, pl1;
        {
            float pp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pp[j] = p[8 + j];
            split_pack<8>(pp, ph1.u, pm1.u, pl1.u);
        }
        AP_WEAVE(12, 4);
        __builtin_amdgcn_sched_barrier(0);
        AP_PV12(vb, ph1, pm1, pl1);
        sc = s + s2;
        __syncthreads();
    }
#undef AP_PV12
#undef AP_VLOAD
#undef AP_WEAVE
#undef AP_QK6
#undef AP_KLOAD
#undef AP_QK_STEP

    if (q < N) {
        const float inv = (l_run > 0.f && q < nq) ? 1.f / l_run : 0.f;
        float *op = O + ((size_t)b * N + q) * ldo + h * AT_D + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(float4 *)(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *(float4 *)(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

// ---- round 5, the default: the same pipelined kernel in the f16x2 arithmetic (split_f16.h) -----------------------------------------------------
// Both operands of both contractions are activations, so there is no packed weight to carry the 2^-11: every operand is split the activation way,
// x -> xh = rne_f16(x), xl = rne_f16((x - xh) 2^11), and a contraction keeps TWO accumulators,
//     main += ah bh          corr += ah bl + al bh          result = main + 2^-11 corr          (dropped: al bl 2^-22, below 2^-24 |a||b|)
// i.e. THREE v_mfma_f32_32x32x16_f16 per block instead of six bf16 ones, 2.5 instead of 5.5 VALU per split element (K, V: once per 256 queries
// at staging; P: per tile in registers), two instead of three term images of K / V in LDS and 32 instead of 48 registers of Q.  The S^T tile already
// had two accumulator chains (s, s2), so "sc = s + s2" becomes one fma per score; O^T gains a correction pair (32 registers).  Every operand
// term is good to 2^-24 relative for |x| >= 2^-12 and to an absolute 2^-36 below (P <= 1: its tiny entries are exact to 2^-36).
#define MFMA_F16(a, b, c) SF_MFMA((a).q, (b).q, (c))
__global__ void __launch_bounds__(512, 1) sg_attention_f16x2_p_kernel(
    const float *__restrict__ Q, const float *__restrict__ Kp, const float *__restrict__ Vp, int ld,
    int N, int heads, int B2, const int *__restrict__ n_tok, int cross, float scale_log2e, float *__restrict__ O, int ldo, int *guard)
{
    __shared__ __attribute__((aligned(16))) unsigned short Ks[2][2][AT_KT][AB_KS];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[2][2][AT_D][AB_VS];
    const int nbh = heads * B2;
    const int bh = blockIdx.x % nbh, qb = blockIdx.x / nbh;
    const int b = bh / heads, h = bh - b * heads;
    const int bk = cross ? (b ^ 1) : b;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int nq = n_tok[b], nk = n_tok[bk];
    const int q0 = qb * (AT_QW * AB_WAVES);
    const int q = q0 + wid * AT_QW + ql;
    if (q0 >= nq) {
        if (q < N) {
            float4 *op = (float4 *)(O + ((size_t)b * N + q) * ldo + h * AT_D + 32 * half);
#pragma unroll
            for (int g = 0; g < 8; ++g) op[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float LS = SF_LOW_SCALE, ILS = 1.0f / SF_LOW_SCALE;

    Frag8 qf[4][2];                                          // [step][h, l]
    {
        const bool ok = q < N;
        const float *qp = Q + ((size_t)b * N + (ok ? q : 0)) * ld + h * AT_D + half * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 t0 = *(const float4 *)(qp + 16 * s), t1 = *(const float4 *)(qp + 16 * s + 4);
            if (!ok) { t0 = make_float4(0.f, 0.f, 0.f, 0.f); t1 = t0; }
            sf_split2(t0.x * scale_log2e, t0.y * scale_log2e, LS, qf[s][0].u[0], qf[s][1].u[0]);
            sf_split2(t0.z * scale_log2e, t0.w * scale_log2e, LS, qf[s][0].u[1], qf[s][1].u[1]);
            sf_split2(t1.x * scale_log2e, t1.y * scale_log2e, LS, qf[s][0].u[2], qf[s][1].u[2]);
            sf_split2(t1.z * scale_log2e, t1.w * scale_log2e, LS, qf[s][0].u[3], qf[s][1].u[3]);
        }
    }
    f32x16 o0, o1, c0, c1;                                   // O^T main (channels 0-31 / 32-63) and correction accumulators
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; c0[r] = 0.f; c1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int sr = tid >> 4, sc4 = (tid & 15) * 4;
    const unsigned rowb = (unsigned)ld * 4u;
    const unsigned span = nk > 0 ? (unsigned)(nk - 1) * rowb + AT_D * 4u : 0u;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(Kp + (size_t)bk * N * ld + h * AT_D), 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(Vp + (size_t)bk * N * ld + h * AT_D), 0, (int)span, 0x00020000);
    const unsigned koff = (unsigned)sr * rowb + 4u * (unsigned)sc4, voff = 4u * (unsigned)lane;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    float4 kr;
    float v0, v1, v2, v3;
    auto gload_k = [&](int t) { kr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rk, koff, (unsigned)(t * AT_KT) * rowb, 0)); };
    auto gload_v = [&](int t) {
        const unsigned sv = (unsigned)(t * AT_KT + 4 * wid) * rowb;
        v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv, 0));
        v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv + rowb, 0));
        v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv + 2u * rowb, 0));
        v3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, sv + 3u * rowb, 0));
    };
    auto lstore_k = [&](int buf) {
        unsigned ph[2], pl[2];
        sf_split2(kr.x, kr.y, LS, ph[0], pl[0]); sf_split2(kr.z, kr.w, LS, ph[1], pl[1]);
        *(uint2 *)&Ks[buf][0][sr][sc4] = make_uint2(ph[0], ph[1]); *(uint2 *)&Ks[buf][1][sr][sc4] = make_uint2(pl[0], pl[1]);
    };
    auto lstore_v = [&](int buf) {
        unsigned ph[2], pl[2];
        sf_split2(v0, v1, LS, ph[0], pl[0]); sf_split2(v2, v3, LS, ph[1], pl[1]);
        const int p0 = 16 * (wid >> 2) + 8 * (wid & 1) + 4 * ((wid >> 1) & 1);
        *(uint2 *)&Vt[buf][0][lane][p0] = make_uint2(ph[0], ph[1]); *(uint2 *)&Vt[buf][1][lane][p0] = make_uint2(pl[0], pl[1]);
    };
    const unsigned short *kq0 = &Ks[0][0][ql][8 * half], *vq0 = &Vt[0][0][ql][8 * half];
    // fragments of S^T's step st (K stage at kq): [h, l]; its three MFMAs: corr += kh ql + kl qh (s2), main += kh qh (s)
#define AF_KLOAD(f, kq, st) do { f[0].q = *(const uint4 *)((kq) + 16 * (st)); f[1].q = *(const uint4 *)((kq) + AT_KT * AB_KS + 16 * (st)); } while (0)
#define AF_QK3(f, st, s, s2) do { s2 = MFMA_F16(f[0], qf[st][1], s2); s = MFMA_F16(f[0], qf[st][0], s); s2 = MFMA_F16(f[1], qf[st][0], s2); } while (0)
    // V^T fragments of 16 keys: f[0] / f[1] = h term of channels 0-31 / 32-63, f[2] / f[3] = l term; six MFMAs
#define AF_VLOAD(f, vq, st) do { \
        f[0].q = *(const uint4 *)((vq) + 16 * (st)); f[1].q = *(const uint4 *)((vq) + 32 * AB_VS + 16 * (st)); \
        f[2].q = *(const uint4 *)((vq) + AT_D * AB_VS + 16 * (st)); f[3].q = *(const uint4 *)((vq) + AT_D * AB_VS + 32 * AB_VS + 16 * (st)); } while (0)
#define AF_PV6(f, ph, pl) do { \
        c0 = MFMA_F16(f[0], pl, c0); c1 = MFMA_F16(f[1], pl, c1); o0 = MFMA_F16(f[0], ph, o0); o1 = MFMA_F16(f[1], ph, o1); \
        c0 = MFMA_F16(f[2], ph, c0); c1 = MFMA_F16(f[3], ph, c1); } while (0)

    f32x16 sc;                                              // the scores of the tile whose softmax is due
    gload_k(0); gload_v(0); lstore_k(0); lstore_v(0);       // (tiles beyond nk: zeros)
    gload_k(1); lstore_k(1);
    gload_k(2); gload_v(1);                                 // staged during iteration 0
    __syncthreads();
    {
        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
        if (ntiles > 0) {
#pragma unroll
            for (int st = 0; st < 4; ++st) { Frag8 kf[2]; AF_KLOAD(kf, kq0, st); AF_QK3(kf, st, s, s2); }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __builtin_fmaf(s2[r], ILS, s[r]);
    }
    __syncthreads();                                        // K stage 0 is rewritten during iteration 0

    // The last iteration multiplies a K stage of zeros (tile ntiles does not exist) and stages tiles that are never used: one basic block per
    // chunk, so that the MFMAs and the VALU work can be woven together, is worth more than the MFMAs it wastes.
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const unsigned short *kq = kq0 + (buf ^ 1) * (2 * AT_KT * AB_KS);
        const unsigned short *vq = vq0 + buf * (2 * AT_D * AB_VS);
        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
        Frag8 ka[2], kb2[2];
        AF_KLOAD(ka, kq, 0);
        AF_KLOAD(kb2, kq, 1);
        const int kb = t * AT_KT + 4 * half;
        if ((t + 1) * AT_KT > nk) {                         // only the last tile can hold keys >= nk (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb + (r & 3) + 8 * (r >> 2) >= nk) sc[r] = -INFINITY;
        }
        // ---- S^T(t+1), step 0  ||  softmax(t): row maximum
        AF_QK3(ka, 0, s, s2);
        AF_KLOAD(ka, kq, 2);
        float mx = sc[0];
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
        mx = fmaxf(mx, sc[15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 1  ||  exponentials of keys 0 .. 7
        AF_QK3(kb2, 1, s, s2);
        AF_KLOAD(kb2, kq, 3);
        float p[16];
        at_f2 rs2; rs2.x = 0.f; rs2.y = 0.f;
        at_f2 mm; mm.x = m_new; mm.y = m_new;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            at_f2 d; d.x = sc[r]; d.y = sc[r + 1];
            d = d - mm;
            at_f2 e; e.x = __builtin_amdgcn_exp2f(d.x); e.y = __builtin_amdgcn_exp2f(d.y);
            rs2 += e;
            p[r] = e.x; p[r + 1] = e.y;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 2  ||  exponentials of keys 8 .. 15
        AF_QK3(ka, 2, s, s2);
#pragma unroll
        for (int r = 8; r < 16; r += 2) {
            at_f2 d; d.x = sc[r]; d.y = sc[r + 1];
            d = d - mm;
            at_f2 e; e.x = __builtin_amdgcn_exp2f(d.x); e.y = __builtin_amdgcn_exp2f(d.y);
            rs2 += e;
            p[r] = e.x; p[r + 1] = e.y;
        }
        float rs = rs2.x + rs2.y;
        rs += __shfl_xor(rs, 32, 64);
        __builtin_amdgcn_sched_barrier(0);
        if (__ballot(m_new != m_run) != 0ull) {             // the running maximum moved for some query of this wavefront: rescale
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; c0[r] *= alpha; c1[r] *= alpha; }
        }
        l_run += rs;
        m_run = m_new;
        // ---- step 3  ||  split of P (first 16 keys), staging of K(t+2) / V(t+1)
        AF_QK3(kb2, 3, s, s2);
        Frag8 va[4];
        AF_VLOAD(va, vq, 0);
        Frag8 ph0, pl0;
#pragma unroll
        for (int j = 0; j < 4; ++j) sf_split2(p[2 * j], p[2 * j + 1], LS, ph0.u[j], pl0.u[j]);
        lstore_k(buf); lstore_v(buf ^ 1);
        gload_k(t + 3); gload_v(t + 2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T P: keys 0 .. 15  ||  split of P (last 16 keys); then keys 16 .. 31
        AF_PV6(va, ph0, pl0);
        Frag8 vb[4];
        AF_VLOAD(vb, vq, 1);
        Frag8 ph1, pl1;
#pragma unroll
        for (int j = 0; j < 4; ++j) sf_split2(p[8 + 2 * j], p[8 + 2 * j + 1], LS, ph1.u[j], pl1.u[j]);
        __builtin_amdgcn_sched_barrier(0);
        AF_PV6(vb, ph1, pl1);
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __builtin_fmaf(s2[r], ILS, s[r]);
        __syncthreads();
    }
#undef AF_PV6
#undef AF_VLOAD
#undef AF_QK3
#undef AF_KLOAD

    if (q < N) {
        const float inv = (l_run > 0.f && q < nq) ? 1.f / l_run : 0.f;
        float *op = O + ((size_t)b * N + q) * ldo + h * AT_D + 4 * half;
        if (guard) {
            // range guard (guard.h): an out-of-range q / k row turns the row's scores, hence its probabilities and its whole output row, into NaN; an
            // out-of-range v[n, d] turns column d of every row into NaN -- so every accumulator is tested, before the normalisation (inv = 0 for a NaN sum)
            float chk = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { MFR_GUARD_ACC(chk, o0[r]); MFR_GUARD_ACC(chk, c0[r]); MFR_GUARD_ACC(chk, o1[r]); MFR_GUARD_ACC(chk, c1[r]); }
            MFR_GUARD_ACC(chk, l_run);
            if (chk != chk) atomicOr(guard, 1);            // (rows >= N of the last block do not reach this point: per-lane test)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = __builtin_fmaf(c0[r], ILS, o0[r]) * inv; o1[r] = __builtin_fmaf(c1[r], ILS, o1[r]) * inv; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(float4 *)(op + 8 * g) = make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
            *(float4 *)(op + 32 + 8 * g) = make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
        }
    }
}

extern "C" {

// q,k,v: [B2, N, ld] fp32 (row = keypoint; channels of head h at [h*64, h*64+64) from the given base
// pointers, so a fused [.., 768] qkv buffer is passed as base, base+256, base+512 with ld = 768).
// out: [B2, N, ldo].  cross != 0: image b attends to image b^1 (the other image of its pair).
// variant: 0 = f16x2, 256 queries per workgroup, score product one tile ahead of the softmax (default); 1 = exact-fp32 matrix instruction
// (128 queries per workgroup); 2 = bf16x3, otherwise as 0
int mfr_sg_attention_variant(const float *q, const float *k, const float *v, int ld, int B2, int N, int heads,
                             const int32_t *n_tok, int cross, float *out, int ldo, int variant, void *stream)
{
    if (!q || !k || !v || !n_tok || !out || B2 <= 0 || N <= 0 || heads <= 0 || (ld & 3) || (ldo & 3)) return MFR_E_ARG;
    if (cross && (B2 & 1)) return MFR_E_ARG;
    if (variant < 0 || variant > 2) return MFR_E_ARG;
    // the buffer descriptors of variants 0 / 2 address an (image, head)'s rows with 32-bit offsets
    if (variant != 1 && (size_t)N * ld * 4 >= 0x7fffffffull) variant = 1;
    const float scale_log2e = 1.4426950408889634f / 8.0f;          // log2(e) / sqrt(64)
    if (variant != 1) {
        const int nqb = (N + AT_QW * AB_WAVES - 1) / (AT_QW * AB_WAVES);
        if (variant == 0)
            hipLaunchKernelGGL(sg_attention_f16x2_p_kernel, dim3(nqb * heads * B2), dim3(512), 0, (hipStream_t)stream, q, k, v, ld, N, heads, B2, n_tok, cross,
                               scale_log2e, out, ldo, mfr_guard_current());
        else
            hipLaunchKernelGGL(sg_attention_bf16x3_p_kernel, dim3(nqb * heads * B2), dim3(512), 0, (hipStream_t)stream, q, k, v, ld, N, heads, B2, n_tok, cross,
                               scale_log2e, out, ldo);
    } else {
        const int nqb = (N + AT_QW * AT_WAVES - 1) / (AT_QW * AT_WAVES);
        hipLaunchKernelGGL(sg_attention_kernel, dim3(nqb * heads * B2), dim3(256), 0, (hipStream_t)stream, q, k, v, ld, N, heads, B2, n_tok, cross,
                           scale_log2e, out, ldo);
    }
    CHECK_LAUNCH();
    return 0;
}

int mfr_sg_attention(const float *q, const float *k, const float *v, int ld, int B2, int N, int heads,
                     const int32_t *n_tok, int cross, float *out, int ldo, void *stream)
{
    return mfr_sg_attention_variant(q, k, v, ld, B2, N, heads, n_tok, cross, out, ldo, 0, stream);
}

}  // extern "C"
