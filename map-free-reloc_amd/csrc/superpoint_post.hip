// superpoint_post.hip -- SuperPoint post-processing on gfx950 (everything after the conv heads).
//
// Reference call site: SuperGlue_matcher (etc/feature_matching_baselines/matchers.py:62-120,
// hyper-parameters :65-71) -> upstream SuperPoint.forward (un-vendored submodule; algorithm per
// SURVEY.md Appendix A.2).  Stages:
//
//   sp_scoremap_kernel   softmax over the 65 detector channels, drop the dustbin, 8x8 pixel-shuffle
//                        [B,65,Hc,Wc] -> [B,8Hc,8Wc]                       (one pass, HBM-bound)
//   sp_nms_kernel        simple_nms(radius r, 2 suppression rounds) fused in ONE pass: the five
//                        (2r+1)^2 max-pools run on an LDS tile with a 5r halo; also applies the
//                        keypoint threshold + border removal and appends survivors to a per-image
//                        candidate list (key = score bits | ~raster index)
//   sp_select_kernel     top-K by (score desc, raster index asc) via 64-bit radix select + bitonic
//                        sort in LDS (or raster order when there are <= K candidates, like upstream's
//                        nonzero() order) -> keypoints (x,y), scores, count
//   sp_sample_kernel     descriptor head output (NHWC, un-normalised) -> per-cell L2 normalise,
//                        bilinear grid_sample(align_corners=True), L2 normalise; one wavefront per
//                        keypoint, 1 KiB coalesced corner reads
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

// ------------------------------------------------------------------------------------------
// one thread per coarse cell; logits NCHW so lanes (adjacent x) read coalesced per channel
__global__ void __launch_bounds__(256) sp_scoremap_kernel(const float *__restrict__ logits, int Hc, int Wc,
                                                          float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= Hc * Wc) return;
    const int y = cell / Wc, x = cell - y * Wc;
    const float *p = logits + (size_t)b * 65 * Hc * Wc + cell;
    float v[65];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 65; ++c) { v[c] = p[(size_t)c * Hc * Wc]; m = fmaxf(m, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 65; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
    const float inv = 1.0f / s;
    const int W8 = Wc * 8;
    float *o = out + (size_t)b * Hc * 8 * W8 + (size_t)(y * 8) * W8 + x * 8;
#pragma unroll
    for (int dy = 0; dy < 8; ++dy) {
        float4 a = make_float4(v[dy * 8] * inv, v[dy * 8 + 1] * inv, v[dy * 8 + 2] * inv, v[dy * 8 + 3] * inv);
        float4 c = make_float4(v[dy * 8 + 4] * inv, v[dy * 8 + 5] * inv, v[dy * 8 + 6] * inv, v[dy * 8 + 7] * inv);
        *(float4 *)(o + (size_t)dy * W8) = a;
        *(float4 *)(o + (size_t)dy * W8 + 4) = c;
    }
}

// ------------------------------------------------------------------------------------------
// fused simple_nms.  Output tile TW x TH, halo 5*R (R = 4 -> 20).
#define NMS_R 4
#define NMS_HALO (5 * NMS_R)
#define NMS_TW 64
#define NMS_TH 32
#define NMS_LW (NMS_TW + 2 * NMS_HALO)   // 104
#define NMS_LH (NMS_TH + 2 * NMS_HALO)   // 72
#define NMS_LS (NMS_LW + 1)               // LDS row stride 105: odd -> lanes walking down a column of strips hit 32 distinct banks
#define NMS_N (NMS_LS * NMS_LH)          // 7560
#define NMS_THREADS 1024                 // 16 wavefronts per CU share the one 117 KiB tile (latency hiding)

// separable (2R+1)^2 max-pool of `src` into `dst` over the whole LDS tile (edges of the tile are
// garbage by construction; each pool consumes R of the halo).  tmp is scratch.
// Each thread produces a strip of 8 consecutive outputs from 16 inputs held in registers
// (log-step sliding maximum: 2 LDS reads and ~5 max per output instead of 9 and 9).
static __device__ __forceinline__ void strip_max9(const float v[16], float o[8])
{
    float m2[15], m4[13], m8[9];
#pragma unroll
    for (int i = 0; i < 15; ++i) m2[i] = fmaxf(v[i], v[i + 1]);
#pragma unroll
    for (int i = 0; i < 13; ++i) m4[i] = fmaxf(m2[i], m2[i + 2]);
#pragma unroll
    for (int i = 0; i < 9; ++i) m8[i] = fmaxf(m4[i], m4[i + 4]);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(m8[i], v[i + 8]);      // window [i, i+8] = outputs centred at i+4
}

// m_in: margin (from the LDS tile's edge) inside which `src` is valid, m_out = m_in + R: margin inside which `dst` is needed.  Round 5: every
// pool works on ITS region only -- the k-th of the five pools is needed R k inside the tile edge and reads what the (k - 1)-th left valid --
// instead of on the whole 104 x 72 tile: 5 768 instead of 9 360 strips per tile, the same values wherever a value is used.
// MASKED (round 6): the pool's input is supp_scores = where(supp_mask, 0, scores) formed ON THE FLY from the scores and the suppression mask's bit rows
// (sup[2 y], sup[2 y + 1] = bits 0-63 / 64-103 of tile row y) -- the masked scores are never materialised.
struct NmsBits { unsigned long long lo, hi; };
static __device__ __forceinline__ unsigned nms_window16(const unsigned long long *rows, int y, int x)       // bits x .. x + 15 of tile row y (x >= 0)
{
    const unsigned long long lo = rows[2 * y], hi = rows[2 * y + 1];
    const unsigned long long v = x >= 64 ? (hi >> (x - 64)) : x == 0 ? lo : ((lo >> x) | (hi << (64 - x)));
    return (unsigned)v & 0xffffu;
}
template <bool MASKED>
static __device__ __forceinline__ void pool9(const float *src, float *tmp, float *dst, int m_in, int m_out, const unsigned long long *sup = nullptr)
{
    // row pass: strip = 8 outputs x0..x0+7 of row y, inputs x0-4..x0+11; rows m_in .. LH - m_in (what the column pass reads)
    // (consecutive lanes -> consecutive rows: stride 105 floats = conflict-free)
    {
        const int st0 = m_out / 8, nst = (NMS_LW - m_out + 7) / 8 - st0, nrows = NMS_LH - 2 * m_in;
        for (int t = threadIdx.x; t < nrows * nst; t += NMS_THREADS) {
            const int st = t / nrows, y = m_in + (t - st * nrows), x0 = (st0 + st) * 8;
            float v[16], o[8];
            const unsigned sb = MASKED ? nms_window16(sup, y, max(x0 - NMS_R, 0)) << (x0 < NMS_R ? NMS_R - x0 : 0) : 0u;      // bit i = suppressed(x0 - 4 + i)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int xx = x0 - NMS_R + i;
                v[i] = (xx >= 0 && xx < NMS_LW) ? src[y * NMS_LS + xx] : -INFINITY;
                if (MASKED && ((sb >> i) & 1u) && v[i] != -INFINITY) v[i] = 0.f;      // out-of-image stays -inf (max_pool2d's padding)
            }
            strip_max9(v, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) tmp[y * NMS_LS + x0 + i] = o[i];
        }
    }
    __syncthreads();
    // column pass IN PLACE (dst == tmp; round 6: one work tile instead of two, 64 KB of LDS per workgroup, TWO workgroups per CU): every strip is read into
    // registers, a barrier, then written -- the strips of a pass are at most NMS_THREADS, one per thread
    {
        const int ys0 = m_out / 8, nys = (NMS_LH - m_out + 7) / 8 - ys0, ncols = NMS_LW - 2 * m_out;
        const int t = threadIdx.x;
        const bool live = t < nys * ncols;                  // (host-side static check: nys * ncols <= NMS_THREADS for every pool)
        const int ys = live ? t / ncols : 0, x = m_out + (live ? t - ys * ncols : 0), y0 = (ys0 + ys) * 8;
        float v[16], o[8];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int yy = y0 - NMS_R + i;
            v[i] = (live && yy >= m_in && yy < NMS_LH - m_in) ? tmp[yy * NMS_LS + x] : -INFINITY;
        }
        strip_max9(v, o);
        __syncthreads();
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (y0 + i < NMS_LH) dst[(y0 + i) * NMS_LS + x] = o[i];
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(NMS_THREADS, 8) sp_nms_kernel(const float *__restrict__ scores, int H, int W, float thr,
                                                     int border, float *__restrict__ nms_out /*may be NULL*/,
                                                     unsigned long long *__restrict__ cand, int cand_cap,
                                                     int *__restrict__ cand_count)
{
    // Round 6: the two MASK pools of simple_nms (max_pool(max_mask) > 0 = a 9 x 9 dilation) run on BIT rows -- 72 rows x 104 bits, a thread per
    // row, a few shifts and ORs -- and the suppressed scores are formed on the fly inside the next pool's row pass: 9 passes over the 104 x 72
    // float tile instead of 15, three float buffers instead of five (rounds 1-5: masks as 0 / 1 floats through the same separable max-pool as the scores).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s = (float *)smem;            // scores (-inf outside the image: max_pool2d padding)
    float *a = s + NMS_N;                // the work tile: row-pass output, then (in place) the pool result
    unsigned long long *mk = (unsigned long long *)(a + NMS_N);      // max_mask bit rows [72][2]
    unsigned long long *hz = mk + 2 * NMS_LH;                        // horizontally dilated rows
    unsigned long long *sup = hz + 2 * NMS_LH;                       // supp_mask bit rows
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * NMS_TW - NMS_HALO, y0 = blockIdx.y * NMS_TH - NMS_HALO;
    const float *img = scores + (size_t)b * H * W;
    for (int i = threadIdx.x; i < NMS_N; i += NMS_THREADS) {
        const int ly = i / NMS_LS, lx = i - ly * NMS_LS;               // lx == NMS_LW is the padding column
        const int gx = x0 + lx, gy = y0 + ly;
        s[i] = (lx < NMS_LW && gx >= 0 && gx < W && gy >= 0 && gy < H) ? img[(size_t)gy * W + gx] : -INFINITY;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr unsigned long long HI_MASK = (1ull << (NMS_LW - 64)) - 1ull;
    // a row of 104 mask bits from a per-pixel predicate: unit = (row, 64-column chunk), one ballot each, wavefront wv takes units wv, wv + 16, ...
    auto dilate = [&]() {                                            // sup = 9 x 9 dilation of mk (bits outside the tile: 0)
        if (threadIdx.x < NMS_LH) {
            const unsigned long long lo = mk[2 * threadIdx.x], hi = mk[2 * threadIdx.x + 1];
            unsigned long long dl = lo, dh = hi;
#pragma unroll
            for (int k = 1; k <= NMS_R; ++k) {
                dl |= (lo << k) | (lo >> k) | (hi << (64 - k));
                dh |= (hi << k) | (hi >> k) | (lo >> (64 - k));
            }
            hz[2 * threadIdx.x] = dl; hz[2 * threadIdx.x + 1] = dh & HI_MASK;
        }
        __syncthreads();
        if (threadIdx.x < NMS_LH) {
            unsigned long long dl = 0ull, dh = 0ull;
#pragma unroll
            for (int k = -NMS_R; k <= NMS_R; ++k) {
                const int yy = (int)threadIdx.x + k;
                if (yy >= 0 && yy < NMS_LH) { dl |= hz[2 * yy]; dh |= hz[2 * yy + 1]; }
            }
            sup[2 * threadIdx.x] = dl; sup[2 * threadIdx.x + 1] = dh;
        }
        __syncthreads();
    };
    pool9<false>(s, a, a, 0, NMS_R);                                 // a = max_pool(scores)
    for (int u = wv; u < 2 * NMS_LH; u += NMS_THREADS / 64) {        // max_mask = scores == max_pool(scores) (in-image cells only)
        const int y = u >> 1, x = 64 * (u & 1) + lane;
        const float sv = x < NMS_LW ? s[y * NMS_LS + x] : -INFINITY;
        const bool bit = x < NMS_LW && sv != -INFINITY && sv == a[y * NMS_LS + x];
        const unsigned long long m = __ballot(bit);
        if (lane == 0) mk[u] = m;
    }
    __syncthreads();
    for (int round = 0; round < 2; ++round) {
        dilate();                                                    // supp_mask = max_pool(max_mask) > 0
        pool9<true>(s, a, a, (2 * round + 2) * NMS_R, (2 * round + 3) * NMS_R, sup);      // a = max_pool(supp_scores), supp_scores = where(supp_mask, 0, scores)
        for (int u = wv; u < 2 * NMS_LH; u += NMS_THREADS / 64) {    // max_mask |= (supp_scores == max_pool(supp_scores)) & ~supp_mask
            const int y = u >> 1, x = 64 * (u & 1) + lane;
            const unsigned long long sm = sup[u];
            const float sv = x < NMS_LW ? s[y * NMS_LS + x] : -INFINITY;
            const float ss = (sv != -INFINITY && ((sm >> lane) & 1ull)) ? 0.f : sv;
            const bool nw = x < NMS_LW && ss == a[y * NMS_LS + x];
            const unsigned long long m = __ballot(nw);
            if (lane == 0) mk[u] |= m & ~sm;
        }
        __syncthreads();
    }
    // emit: where(max_mask, scores, 0); candidates above threshold and inside the border.
    // Survivors are first compacted in LDS (the pooling scratch is dead by now) and the workgroup
    // reserves its range of the per-image list with ONE global atomic: per-candidate atomics on the
    // 32 adjacent per-image counters (one cache line) serialised the whole launch (1.7 ms -> ...).
    unsigned long long *lkeys = (unsigned long long *)a;          // <= NMS_TW*NMS_TH keys = 16 KB
    int *lcount = (int *)hz, *lbase = lcount + 1;
    if (threadIdx.x == 0) *lcount = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < NMS_TW * NMS_TH; i += NMS_THREADS) {
        const int ty = i / NMS_TW, tx = i - ty * NMS_TW;
        const int gx = blockIdx.x * NMS_TW + tx, gy = blockIdx.y * NMS_TH + ty;
        if (gx >= W || gy >= H) continue;
        const int li = (ty + NMS_HALO) * NMS_LS + tx + NMS_HALO;
        const int mx = tx + NMS_HALO;
        const float v = ((mk[2 * (ty + NMS_HALO) + (mx >> 6)] >> (mx & 63)) & 1ull) ? s[li] : 0.f;
        if (nms_out) nms_out[(size_t)b * H * W + (size_t)gy * W + gx] = v;
        if (v > thr && gx >= border && gx < W - border && gy >= border && gy < H - border) {
            const int slot = atomicAdd(lcount, 1);
            const unsigned idx = (unsigned)(gy * W + gx);
            lkeys[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(~idx);
        }
    }
    __syncthreads();
    const int nloc = *lcount;
    if (threadIdx.x == 0) *lbase = nloc ? atomicAdd(&cand_count[b], nloc) : 0;
    __syncthreads();
    const int base = *lbase;
    for (int i = threadIdx.x; i < nloc; i += NMS_THREADS)
        if (base + i < cand_cap) cand[(size_t)b * cand_cap + base + i] = lkeys[i];
}

// ------------------------------------------------------------------------------------------
// one 1024-thread workgroup per image.  keys are unique (raster index in the low word).
#define SEL_T 1024
__global__ void __launch_bounds__(SEL_T) sp_select_kernel(const unsigned long long *__restrict__ cand, int cand_cap,
                                                          const int *__restrict__ cand_count, int W, int K,
                                                          float *__restrict__ kpts /*[B,K,2]*/,
                                                          float *__restrict__ kscores /*[B,K]*/,
                                                          int *__restrict__ n_kpts /*[B]*/)
{
    __shared__ unsigned long long sel[SEL_T];
    __shared__ int hist[256];
    __shared__ unsigned long long prefix_s;
    __shared__ int remaining_s, nsel_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = cand_count[b];
    if (n > cand_cap) n = cand_cap;
    const unsigned long long *c = cand + (size_t)b * cand_cap;
    const bool topk = n > K;
    unsigned long long kth = 0ull;         // smallest selected key when topk
    if (topk) {
        // radix select of the K-th largest key, 8 bits per pass from the top
        unsigned long long prefix = 0ull, pmask = 0ull;
        int remaining = K;
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += SEL_T) {
                const unsigned long long k = c[i];
                if ((k & pmask) == prefix) atomicAdd(&hist[(int)((k >> shift) & 0xff)], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0, d = 255;
                for (; d >= 0; --d) {
                    if (acc + hist[d] >= remaining) break;
                    acc += hist[d];
                }
                prefix_s = prefix | ((unsigned long long)d << shift);
                remaining_s = remaining - acc;
            }
            __syncthreads();
            prefix = prefix_s; remaining = remaining_s;
            pmask |= 0xffull << shift;
            __syncthreads();
        }
        kth = prefix;
    }
    if (tid == 0) nsel_s = 0;
    sel[tid] = 0ull;
    __syncthreads();
    for (int i = tid; i < n; i += SEL_T) {
        const unsigned long long k = c[i];
        if (!topk || k >= kth) {
            const int slot = atomicAdd(&nsel_s, 1);
            if (slot < SEL_T) sel[slot] = k;
        }
    }
    __syncthreads();
    const int nsel = min(nsel_s, min(K, SEL_T));
    // sort key: top-K -> (score desc, index asc) == key desc ; else raster order == low word desc
    // (low word is ~index).  Unused slots hold 0 and sink to the end.
    unsigned long long mine = sel[tid];
    if (!topk && tid < nsel_s) mine = ((mine & 0xffffffffull) << 32) | (mine >> 32);
    sel[tid] = mine;
    __syncthreads();
    for (int k2 = 2; k2 <= SEL_T; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const int ixj = tid ^ j;
            if (ixj > tid) {
                const unsigned long long x = sel[tid], y = sel[ixj];
                const bool desc = ((tid & k2) == 0);
                if ((x < y) == desc) { sel[tid] = y; sel[ixj] = x; }
            }
            __syncthreads();
        }
    }
    if (tid < K) {
        float kx = 0.f, ky = 0.f, sc = 0.f;
        if (tid < nsel) {
            unsigned long long k = sel[tid];
            if (!topk) k = ((k & 0xffffffffull) << 32) | (k >> 32);
            const unsigned idx = ~(unsigned)(k & 0xffffffffull);
            sc = __uint_as_float((unsigned)(k >> 32));
            kx = (float)(idx % (unsigned)W); ky = (float)(idx / (unsigned)W);
        }
        kpts[((size_t)b * K + tid) * 2] = kx;
        kpts[((size_t)b * K + tid) * 2 + 1] = ky;
        kscores[(size_t)b * K + tid] = sc;
    }
    if (tid == 0) n_kpts[b] = nsel;
}

// ------------------------------------------------------------------------------------------
// one wavefront per keypoint; dense [B,Hc,Wc,256] NHWC raw descriptors
__global__ void __launch_bounds__(256) sp_sample_kernel(const float *__restrict__ dense, int Hc, int Wc,
                                                        const float *__restrict__ kpts, const int *__restrict__ n_kpts,
                                                        int K, float *__restrict__ desc /*[B,K,256]*/)
{
    const int b = blockIdx.y;
    const int kp = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (kp >= K) return;
    float4 *o = (float4 *)(desc + ((size_t)b * K + kp) * 256) + lane;
    if (kp >= n_kpts[b]) { *o = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float s = 8.f;
    float kx = kpts[((size_t)b * K + kp) * 2], ky = kpts[((size_t)b * K + kp) * 2 + 1];
    // upstream sample_descriptors: (k - s/2 + 0.5) / (W*s - s/2 - 0.5) * 2 - 1, then
    // grid_sample align_corners=True: ix = (g + 1) / 2 * (Wc - 1)
    float gx = (kx - s / 2 + 0.5f) / ((float)Wc * s - s / 2 - 0.5f) * 2.f - 1.f;
    float gy = (ky - s / 2 + 0.5f) / ((float)Hc * s - s / 2 - 0.5f) * 2.f - 1.f;
    const float ix = (gx + 1.f) * 0.5f * (float)(Wc - 1), iy = (gy + 1.f) * 0.5f * (float)(Hc - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const int xs[4] = { x0, x1, x0, x1 }, ys[4] = { y0, y0, y1, y1 };
    const float ws[4] = { wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1 };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (xs[c] < 0 || xs[c] >= Wc || ys[c] < 0 || ys[c] >= Hc) continue;       // zero padding
        const float4 v = *((const float4 *)(dense + (((size_t)b * Hc + ys[c]) * Wc + xs[c]) * 256) + lane);
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
        const float inv = ws[c] / fmaxf(sqrtf(ss), 1e-12f);                        // F.normalize eps
        acc.x += v.x * inv; acc.y += v.y * inv; acc.z += v.z * inv; acc.w += v.w * inv;
    }
    float ss = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    *o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

extern "C" {

int mfr_sp_scoremap(const float *logits, int B, int Hc, int Wc, float *scores, void *stream)
{
    if (!logits || !scores || B <= 0 || Hc <= 0 || Wc <= 0) return MFR_E_ARG;
    hipLaunchKernelGGL(sp_scoremap_kernel, dim3((Hc * Wc + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       logits, Hc, Wc, scores);
    CHECK_LAUNCH();
    return 0;
}

int mfr_sp_nms_candidates(const float *scores, int B, int H, int W, int nms_radius, float threshold, int border,
                          float *nms_out, uint64_t *cand, int cand_cap, int32_t *cand_count, void *stream)
{
    if (!scores || !cand || !cand_count || B <= 0 || H <= 0 || W <= 0 || cand_cap <= 0) return MFR_E_ARG;
    if (nms_radius != NMS_R) return MFR_E_ARG;          // compiled for the reference's radius (matchers.py:65)
    hipStream_t s = (hipStream_t)stream;
    if (mfr_zero_async(cand_count, sizeof(int32_t) * (size_t)B, s) != hipSuccess) return MFR_E_LAUNCH;
    dim3 grid((W + NMS_TW - 1) / NMS_TW, (H + NMS_TH - 1) / NMS_TH, B);
    // 62.5 KiB of LDS (two float tiles + three sets of bit rows): two workgroups per CU
    const size_t smem = 2 * NMS_N * sizeof(float) + 3 * 2 * NMS_LH * sizeof(unsigned long long);
    if (hipFuncSetAttribute((const void *)sp_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return MFR_E_LAUNCH;
    hipLaunchKernelGGL(sp_nms_kernel, grid, dim3(NMS_THREADS), smem, s, scores, H, W, threshold, border,
                       nms_out, (unsigned long long *)cand, cand_cap, cand_count);
    CHECK_LAUNCH();
    return 0;
}

int mfr_sp_select_topk(const uint64_t *cand, int cand_cap, const int32_t *cand_count, int B, int W, int K,
                       float *kpts, float *kscores, int32_t *n_kpts, void *stream)
{
    if (!cand || !cand_count || !kpts || !kscores || !n_kpts || B <= 0 || K <= 0 || K > SEL_T || W <= 0)
        return MFR_E_ARG;
    hipLaunchKernelGGL(sp_select_kernel, dim3(B), dim3(SEL_T), 0, (hipStream_t)stream,
                       (const unsigned long long *)cand, cand_cap, cand_count, W, K, kpts, kscores, n_kpts);
    CHECK_LAUNCH();
    return 0;
}

int mfr_sp_sample_descriptors(const float *dense_nhwc, int B, int Hc, int Wc, const float *kpts,
                              const int32_t *n_kpts, int K, float *desc, void *stream)
{
    if (!dense_nhwc || !kpts || !n_kpts || !desc || B <= 0 || Hc <= 0 || Wc <= 0 || K <= 0) return MFR_E_ARG;
    hipLaunchKernelGGL(sp_sample_kernel, dim3((K + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, dense_nhwc, Hc, Wc,
                       kpts, n_kpts, K, desc);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
