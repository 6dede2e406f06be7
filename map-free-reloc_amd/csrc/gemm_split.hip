// gemm_split.hip -- Y[M, N] (+)= act(X[M, K] W[N, K]^T + bias): the linear layers of the SuperGlue / LoFTR transformers (fp32 in,
// fp32 out) on the gfx950 16-bit matrix cores at fp32 accuracy, by operand splitting.
//
// Reference call site: SuperGlue_matcher / LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-120) -> the un-vendored
// networks' Conv1d(k=1) / Linear layers; rounds 1-2 ran them as library (hipBLASLt) fp32 GEMMs.
//
// Two arithmetics, one kernel structure (template parameter F16):
//   bf16x3 (rounds 3-4)  every fp32 operand is split EXACTLY into three bf16 terms x = h + m + l (truncation, 8 + 8 + 8 significand bits); a
//                        product is the six partial products hh + hm + mh + hl + lh + mm (each exact in fp32) accumulated in fp32 by
//                        v_mfma_f32_32x32x16_bf16.  Error = that of the exact-fp32 matrix instruction (profiles/r03_bf16x3_probe.jsonl).
//   f16x2  (round 5, the default)  split_f16.h: activations as two f16 terms (the low one scaled by 2^11), weights pre-scaled per output
//                        feature and packed as two f16 terms (wh, wl; the third, wq = wh 2^-11, is one v_pk_mul_f16 per four weights on the
//                        fragment a wavefront has just read -- exact, and a third less W traffic through L2 and LDS than carrying it):
//                        THREE v_mfma_f32_32x32x16_f16 per K block, 2.5 instead of 5.5 VALU per activation element, two instead of three
//                        term images of X AND of W in LDS.  Same error class for
//                        2^-14 <= |x| <= 65504 (profiles/r05_f16x2_probe.jsonl); the epilogue multiplies by the feature's 1 / scale.
//
// Mapping (all kernels of this file): workgroup tile = 128 rows x 128 output features, 4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles
// (64 accumulator registers); K in steps of 32.  W is split and packed ONCE per weight set in the exact image a workgroup stages (three
// terms x [k group of 8][feature][8 x 16 bit]); X is read as fp32 (coalesced 128-byte row pieces), split by the staging threads -- each
// element once per workgroup -- and written to LDS in the same fragment order, so every MFMA operand is one conflict-free ds_read_b128.
// Output features run along the lanes: 128-byte stores.
//
// Kernels; every one sums each output element in the same order (bitwise-equal results per arithmetic, tests/test_gpu_gemm_split.py):
//   gemm_split_kernel      one tile per workgroup, register-staged prefetch (round 3; flag 4: the baseline of the bitwise test)
//   gemm_split_pk_kernel   persistent workgroups walking XCD-local tile lists (flag 8); runs K % 64 != 0
//   gemm_split_d_kernel    persistent, W by LDS-DMA into two W stages, X two K steps ahead -- the default (K % 64 == 0)
// (Round 4's other generations -- deferred tile stores, the eight-wavefront 256 x 128 kernel, the timing ablations -- measured no faster
// and left the library in round 5; their measurements are profiles/r04_ablate_gemm.json.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "split_f16.h"
#include "guard.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB_BM 128
#define GB_BN 128
#define GB_BK 32
#define GB_KG_STRIDE 129                  // 16-byte units per k group (128 rows + 1 pad: conflict-free stores)
#define GB_TERM_UNITS (4 * GB_KG_STRIDE)  // units per X term image
#define GB_WT(F16) ((F16) ? 2 : 3)        // W terms STORED per tile: bf16x3 h, m, l; f16x2 wh, wl (wq = wh 2^-11 is formed in registers, below)
#define GB_W_TILE_UNITS(F16) (GB_WT(F16) * 512)   // packed W tile: terms x 4 k groups x 128 features, 16 bytes each

union GbFrag { bf16x8 v; unsigned u[4]; uint4 q; };

__device__ __forceinline__ void gb_split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned gb_pack(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// f16x2: wq = rne_f16(wh * 2^-11), eight weights at a time (4 x v_pk_mul_f16; a power of two: exact unless the product is subnormal)
typedef _Float16 gb_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint4 gb_wq(const uint4 &wh)
{
    const gb_h8 h = __builtin_bit_cast(gb_h8, wh);
    const gb_h8 q = h * (_Float16)(1.0f / SF_LOW_SCALE);
    return __builtin_bit_cast(uint4, q);
}

// eight consecutive K elements of one row -> the row's 16-byte unit of every X term image (XT = 3: h, m, l bf16; XT = 2: xh, xl f16)
template <bool F16>
__device__ __forceinline__ void gb_xsplit_store(uint4 *lds, const float4 &p, const float4 &q, int dst)
{
    if (F16) {
        unsigned h[4], l[4];
        sf_split2(p.x, p.y, SF_LOW_SCALE, h[0], l[0]); sf_split2(p.z, p.w, SF_LOW_SCALE, h[1], l[1]);
        sf_split2(q.x, q.y, SF_LOW_SCALE, h[2], l[2]); sf_split2(q.z, q.w, SF_LOW_SCALE, h[3], l[3]);
        lds[0 * GB_TERM_UNITS + dst] = make_uint4(h[0], h[1], h[2], h[3]);
        lds[1 * GB_TERM_UNITS + dst] = make_uint4(l[0], l[1], l[2], l[3]);
    } else {
        const float x[8] = { p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w };
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gb_split3(x[e], h[e], m[e], l[e]);
        lds[0 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(h[0], h[1]), gb_pack(h[2], h[3]), gb_pack(h[4], h[5]), gb_pack(h[6], h[7]));
        lds[1 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(m[0], m[1]), gb_pack(m[2], m[3]), gb_pack(m[4], m[5]), gb_pack(m[6], m[7]));
        lds[2 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(l[0], l[1]), gb_pack(l[2], l[3]), gb_pack(l[4], l[5]), gb_pack(l[6], l[7]));
    }
}

// the partial products of one 16-wide K block into the 2 x 2 accumulator tiles, small terms first; a: X terms, b: W terms
#define GB_MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
template <bool F16>
__device__ __forceinline__ void gb_products(f32x16 (&acc)[2][2], const GbFrag (&a)[2][3], const GbFrag (&b)[2][3])
{
#define GB_P4(ta, tb) do { \
        if (F16) { acc[0][0] = SF_MFMA(a[0][ta].q, b[0][tb].q, acc[0][0]); acc[0][1] = SF_MFMA(a[0][ta].q, b[1][tb].q, acc[0][1]); \
                   acc[1][0] = SF_MFMA(a[1][ta].q, b[0][tb].q, acc[1][0]); acc[1][1] = SF_MFMA(a[1][ta].q, b[1][tb].q, acc[1][1]); } \
        else     { acc[0][0] = GB_MFMA_BF(a[0][ta].v, b[0][tb].v, acc[0][0]); acc[0][1] = GB_MFMA_BF(a[0][ta].v, b[1][tb].v, acc[0][1]); \
                   acc[1][0] = GB_MFMA_BF(a[1][ta].v, b[0][tb].v, acc[1][0]); acc[1][1] = GB_MFMA_BF(a[1][ta].v, b[1][tb].v, acc[1][1]); } } while (0)
    if (F16) { GB_P4(1, 2); GB_P4(0, 1); GB_P4(0, 0); }                              // xl wq, xh wl, xh wh
    else     { GB_P4(1, 1); GB_P4(0, 2); GB_P4(2, 0); GB_P4(0, 1); GB_P4(1, 0); GB_P4(0, 0); }
#undef GB_P4
}
#define GB_XT(F16) ((F16) ? 2 : 3)
// batched launches (gridDim.y problems that share shapes: the matchers' score / similarity products, one packed "weight" per pair): element
// strides of X and Y, 16-byte-unit stride of the packed operand (its 1 / scale tail moves with it)
#define GB_BATCH_OFFSETS() do { const size_t bz_ = blockIdx.y; X += bz_ * (size_t)xs; Wp += bz_ * (size_t)ws; Y += bz_ * (size_t)ys; if (F16) oscale += bz_ * (size_t)ws * 4; } while (0)

// ---- weight packing ----------------------------------------------------------------------------------------------------------------------
// W [N, K] f32 row-major -> packed [n block][k block][term][k group][feature (128)][8 x 16 bit]; one thread per 16-byte unit.
// f16x2: the blob ends with the per-feature 1 / scale (nnb x 128 floats, written by gb_scale_kernel BEFORE this kernel runs).
__global__ void __launch_bounds__(64) gb_scale_kernel(const float *__restrict__ w, int N, int K, int npad, float *__restrict__ oscale, float out_mul,
                                                      int ldw, long long w_stride, long long p_stride_f)
{
    w += (size_t)blockIdx.y * (size_t)w_stride; oscale += (size_t)blockIdx.y * (size_t)p_stride_f;
    const int n = blockIdx.x, lane = threadIdx.x;
    float mx = 0.f;
    if (n < N)
        for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(w[(size_t)n * ldw + k]));
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0 && n < npad) oscale[n] = out_mul / sf_feature_scale(mx);           // powers of two: the reciprocal is exact (out_mul: a power of two the caller wants folded into the product)
}
template <bool F16>
__global__ void __launch_bounds__(256) gb_pack_kernel(const float *__restrict__ w, int N, int K, long long total, const float *__restrict__ oscale, uint4 *__restrict__ out,
                                                      float out_mul, int ldw, long long w_stride, long long p_stride_u)
{
    w += (size_t)blockIdx.y * (size_t)w_stride; out += (size_t)blockIdx.y * (size_t)p_stride_u;
    if (F16) oscale += (size_t)blockIdx.y * (size_t)p_stride_u * 4;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int u = (int)(t % GB_W_TILE_UNITS(F16));
    const long long tile = t / GB_W_TILE_UNITS(F16);
    const int nkb = K / GB_BK;
    const int kb = (int)(tile % nkb), nb = (int)(tile / nkb);
    const int term = u / 512, kg = (u % 512) / 128, f = u % 128;
    const int n = nb * GB_BN + f, k0 = kb * GB_BK + 8 * kg;
    unsigned word[8];
    const float s = F16 ? out_mul / oscale[n] : 1.0f;            // (oscale = out_mul / scale)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (n < N) ? w[(size_t)n * ldw + k0 + e] : 0.f;
        if (F16) {
            unsigned short wh, wl, wq;
            sf_split_w(x * s, wh, wl, wq);
            (void)wq;
            word[e] = (unsigned)(term == 0 ? wh : wl) << 16;                         // (upper half, as the bf16 terms: gb_pack takes the upper halves)
        } else {
            unsigned h, m, l;
            gb_split3(x, h, m, l);
            word[e] = term == 0 ? h : term == 1 ? m : l;
        }
    }
    out[t] = make_uint4(gb_pack(word[0], word[1]), gb_pack(word[2], word[3]), gb_pack(word[4], word[5]), gb_pack(word[6], word[7]));
}

// ---- baseline: one tile per workgroup (round 3) --------------------------------------------------------------------------------------
// FLAGS: 1 = ReLU, 2 = accumulate into Y (Y += ...)
template <int FLAGS, bool F16>
__global__ void __launch_bounds__(256, 2) gemm_split_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, const float *__restrict__ oscale,
                                                            const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K, int nnb,
                                                            long long xs, long long ws, long long ys, int *guard)
{
    GB_BATCH_OFFSETS();
    constexpr int XT = GB_XT(F16);
    constexpr int WT = GB_WT(F16);
    __shared__ uint4 lds[(XT + WT) * GB_TERM_UNITS];      // X terms, W terms
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // feature blocks innermost: the workgroups that share an X row block run back to back (its tiles stay in L2)
    const int nb = blockIdx.x % nnb, mb = blockIdx.x / nnb;
    const int m0 = mb * GB_BM;
    const int nkb = K / GB_BK;

    // staging assignment.  X: unit u = tid + 256 i -> (row = u >> 2, k group = u & 3), 8 floats = two 16-byte loads.
    // W: unit u = tid + 256 i, i = 0..5 -> straight copy of the packed tile image.
    const float *xrow[2];
    int xdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 256 * i, row = u >> 2, kg = u & 3;
        const int m = min(m0 + row, M - 1);               // rows beyond M: a valid row is read and its results are never stored
        xrow[i] = X + (size_t)m * ldx + 8 * kg;
        xdst[i] = kg * GB_KG_STRIDE + row;
    }
    int wdst[2 * WT];
#pragma unroll
    for (int i = 0; i < 2 * WT; ++i) {
        const int u = tid + 256 * i, term = u / 512, kg = (u % 512) / 128, f = u % 128;
        wdst[i] = (XT + term) * GB_TERM_UNITS + kg * GB_KG_STRIDE + f;
    }
    const uint4 *wtile = Wp + (size_t)nb * nkb * GB_W_TILE_UNITS(F16) + tid;

    // (named registers, not arrays: hipcc keeps a lambda-captured array that is written under a condition in scratch memory)
    float4 xa0, xa1, xb0, xb1;
    uint4 w0, w1, w2, w3, w4, w5;
#define GB_GLOAD(kb) do { \
        xa0 = *(const float4 *)(xrow[0] + (kb) * GB_BK); xa1 = *(const float4 *)(xrow[0] + (kb) * GB_BK + 4); \
        xb0 = *(const float4 *)(xrow[1] + (kb) * GB_BK); xb1 = *(const float4 *)(xrow[1] + (kb) * GB_BK + 4); \
        const uint4 *wt_ = wtile + (size_t)(kb) * GB_W_TILE_UNITS(F16); \
        w0 = wt_[0]; w1 = wt_[256]; w2 = wt_[512]; w3 = wt_[768]; if (WT == 3) { w4 = wt_[1024]; w5 = wt_[1280]; } } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: operand row (token / feature) = 64 w + 32 t + (lane & 31), k group = 2 ks + (lane >> 5)
    const int arow = (lane >> 5) * GB_KG_STRIDE + 64 * wm + (lane & 31);
    const int brow = (lane >> 5) * GB_KG_STRIDE + 64 * wn + (lane & 31);

    GB_GLOAD(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();                                  // the previous step's fragment reads are done
        gb_xsplit_store<F16>(lds, xa0, xa1, xdst[0]); gb_xsplit_store<F16>(lds, xb0, xb1, xdst[1]);
        lds[wdst[0]] = w0; lds[wdst[1]] = w1; lds[wdst[2]] = w2; lds[wdst[3]] = w3;
        if (WT == 3) { lds[wdst[2 * WT - 2]] = w4; lds[wdst[2 * WT - 1]] = w5; }
        __syncthreads();
        { const int kn = min(kb + 1, nkb - 1); GB_GLOAD(kn); }   // in flight during the MFMAs below (the last step re-reads its own tile)
        // without this fence hipcc sinks the loads BELOW the MFMAs (ten live 16-byte registers fewer across them) and every K step
        // pays the full memory latency before its split: load -> wait -> split -> store -> MFMA, nothing overlapped inside a workgroup
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GbFrag a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int t = 0; t < XT; ++t) a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i];
#pragma unroll
                for (int t = 0; t < WT; ++t) b[i][t].q = lds[(XT + t) * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + brow + 32 * i];
                if (F16) b[i][2].q = gb_wq(b[i][0].q);
            }
            gb_products<F16>(acc, a, b);
        }
    }
#undef GB_GLOAD

    if (F16 && guard) {                                     // range guard (guard.h): one column block covers the tile's rows
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) MFR_GUARD_ACC(chk, acc[i][0][r]);
        mfr_guard_commit(guard, chk);
    }
    // epilogue: accumulator register r of tile (i, j): token row 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), feature 64 wn + 32 j + (lane & 31)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nb * GB_BN + 64 * wn + 32 * j + (lane & 31);
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
        const float os = F16 ? oscale[n] : 1.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                float *yp = Y + (size_t)m * ldy + n;
                float v = F16 ? __builtin_fmaf(acc[i][j][r], os, bv) : acc[i][j][r] + bv;
                if (FLAGS & 1) v = fmaxf(v, 0.f);
                if (FLAGS & 2) v += *yp;
                *yp = v;
            }
        }
    }
}


// ---- round 4: PERSISTENT workgroups -------------------------------------------------------------------------------------------------------
// Measured on the SuperGlue shapes (M = 65536; profiles/r04_bench_sg_pnp_kernel_stats.csv) the one-tile-per-workgroup kernel above takes
// matrix-core time + HBM time + ~30 us, not their maximum: all resident workgroups start together, run the same K loop and store their
// 64 KB tiles together, so the matrix cores idle while 50 MB of tiles drain and HBM idles while they multiply; a new workgroup waits for the
// previous one's stores and then for its own first loads.  Here a workgroup walks over tiles: the loads of the next tile's first K step are
// in flight during the current tile's last multiply and -- for the accumulating epilogue -- Y is fetched during the last K step.
// Same arithmetic per output element as the kernel above, bit for bit.  Grid = 2 workgroups per CU (512: the SuperGlue shapes are
// 1024 / 2048 / 3072 tiles); tiles are dealt per XCD so that the workgroups sharing an X row block share an L2 (row block mb lives on XCD mb % 8).
#define GB_RSRC_FLAGS 0x00020000
template <int FLAGS, bool F16>
__global__ void __launch_bounds__(256, 2) gemm_split_pk_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, const float *__restrict__ oscale,
                                                               const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K, int nnb, int nmb,
                                                               long long xs, long long ws, long long ys, int *guard)
{
    GB_BATCH_OFFSETS();
    constexpr int XT = GB_XT(F16);
    constexpr int WT = GB_WT(F16);
    __shared__ uint4 lds[(XT + WT) * GB_TERM_UNITS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int nkb = K / GB_BK;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int items = ((nmb + 7) >> 3) * nnb;               // work items of one XCD: (local row block, feature block), feature block innermost
    // item j of this XCD -> tile; row blocks beyond nmb do not exist (they can only be the last local row block: the walk ends there)
#define GB_TILE(j, mb_, nb_) const int nb_ = (j) % nnb, mb_ = ((j) / nnb) * 8 + xcd
    int j = slot;
    if (j >= items) return;
    { GB_TILE(j, mb, nb); (void)nb; if (mb >= nmb) return; }

    int xdst[2], xr[2], xk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 256 * i;
        xr[i] = u >> 2; xk[i] = 8 * (u & 3);
        xdst[i] = (u & 3) * GB_KG_STRIDE + xr[i];
    }
    int wdst[2 * WT];
#pragma unroll
    for (int i = 0; i < 2 * WT; ++i) {
        const int u = tid + 256 * i, term = u / 512, kg = (u % 512) / 128, f = u % 128;
        wdst[i] = (XT + term) * GB_TERM_UNITS + kg * GB_KG_STRIDE + f;
    }
    const int arow = (lane >> 5) * GB_KG_STRIDE + 64 * wm + (lane & 31);
    const int brow = (lane >> 5) * GB_KG_STRIDE + 64 * wn + (lane & 31);
    const unsigned rowb = (unsigned)ldy * 4u;              // bytes per row of Y
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, bias ? N * 4 : 0, GB_RSRC_FLAGS);   // no bias: every read returns 0
    const __amdgpu_buffer_rsrc_t rscale = __builtin_amdgcn_make_buffer_rsrc((void *)oscale, 0, F16 ? nnb * GB_BN * 4 : 0, GB_RSRC_FLAGS);

    // the LOAD stream runs one K step ahead of the multiply, across tile boundaries
    const float *lx0, *lx1;
    const uint4 *lw;
    int lk, lj = j;                                        // K step / item the load stream is at
    auto load_tile_start = [&](int jj) {
        GB_TILE(jj, mb, nb);
        lx0 = X + (size_t)min(mb * GB_BM + xr[0], M - 1) * ldx + xk[0];      // rows beyond M: a valid row is read, its results are never stored
        lx1 = X + (size_t)min(mb * GB_BM + xr[1], M - 1) * ldx + xk[1];
        lw = Wp + (size_t)nb * nkb * GB_W_TILE_UNITS(F16) + tid;
        lk = 0;
    };
    float4 xa0, xa1, xb0, xb1;
    uint4 w0, w1, w2, w3, w4, w5;
#define GB_PLOAD() do { \
        xa0 = *(const float4 *)lx0; xa1 = *(const float4 *)(lx0 + 4); xb0 = *(const float4 *)lx1; xb1 = *(const float4 *)(lx1 + 4); \
        w0 = lw[0]; w1 = lw[256]; w2 = lw[512]; w3 = lw[768]; if (WT == 3) { w4 = lw[1024]; w5 = lw[1280]; } \
        lx0 += GB_BK; lx1 += GB_BK; lw += GB_W_TILE_UNITS(F16); \
        if (++lk == nkb) { int jn = lj + per_xcd; if (jn < items) { GB_TILE(jn, mbn_, nbn_); (void)nbn_; if (mbn_ >= nmb) jn = items; } \
                           if (jn < items) lj = jn; load_tile_start(lj); } } while (0)        /* no next tile: the last load re-reads this tile's start */

    f32x16 acc[2][2];
    unsigned ov[2][2][16];                                 // FLAGS & 2: the tile of Y fetched during the last K step
    // accumulator register r of tile (i, jj) of a wavefront: row 32 i + (r & 3) + 8 (r >> 2) (+ 64 wm + 4 (lane >> 5): the lane offset), feature + 32 jj
#define GB_SOFF(i, r) ((unsigned)(32 * (i) + ((r) & 3) + 8 * ((r) >> 2)) * rowb)

    load_tile_start(j);
    GB_PLOAD();
    for (;;) {
        GB_TILE(j, mb, nb);
        const int m0 = mb * GB_BM;
        // this tile's rows of Y as one buffer (rows beyond M fall outside it: reads return 0, stores are dropped)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(Y + (size_t)m0 * ldy), 0, (int)((unsigned)min(GB_BM, M - m0) * rowb), GB_RSRC_FLAGS);
        const int n0 = nb * GB_BN + 64 * wn + (lane & 31);
        const unsigned yoff = (unsigned)(64 * wm + 4 * (lane >> 5)) * rowb + 4u * (unsigned)n0;
        const float bv0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0, 0, 0));
        const float bv1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0 + 128u, 0, 0));
        const float os0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rscale, 4u * (unsigned)n0, 0, 0));
        const float os1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rscale, 4u * (unsigned)n0 + 128u, 0, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
#define GB_PSTEP(LAST) do { \
            __syncthreads(); \
            gb_xsplit_store<F16>(lds, xa0, xa1, xdst[0]); gb_xsplit_store<F16>(lds, xb0, xb1, xdst[1]); \
            lds[wdst[0]] = w0; lds[wdst[1]] = w1; lds[wdst[2]] = w2; lds[wdst[3]] = w3; \
            if (WT == 3) { lds[wdst[2 * WT - 2]] = w4; lds[wdst[2 * WT - 1]] = w5; } \
            __syncthreads(); \
            GB_PLOAD(); \
            if ((LAST) && (FLAGS & 2)) { \
                _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) \
                            ov[i][jj][r] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff + 128u * jj, GB_SOFF(i, r), 0); } \
            __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
                GbFrag a[2][3], b[2][3]; \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
                    _Pragma("unroll") for (int t = 0; t < XT; ++t) a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i]; \
                    _Pragma("unroll") for (int t = 0; t < WT; ++t) b[i][t].q = lds[(XT + t) * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + brow + 32 * i]; \
                    if (F16) b[i][2].q = gb_wq(b[i][0].q); } \
                gb_products<F16>(acc, a, b); } } while (0)
        for (int kb = 0; kb < nkb - 1; ++kb) GB_PSTEP(false);
        GB_PSTEP(true);

        if (F16 && guard) {                                 // range guard (guard.h): one column block covers the tile's rows
            float chk = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) MFR_GUARD_ACC(chk, acc[i][0][r]);
            mfr_guard_commit(guard, chk);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (n0 + 32 * jj >= N) continue;
            const float bv = jj ? bv1 : bv0, os = jj ? os1 : os0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = F16 ? __builtin_fmaf(acc[i][jj][r], os, bv) : acc[i][jj][r] + bv;
                    if (FLAGS & 1) v = fmaxf(v, 0.f);
                    if (FLAGS & 2) v += __builtin_bit_cast(float, ov[i][jj][r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, yoff + 128u * jj, GB_SOFF(i, r), 0);
                }
        }
        // next item of this workgroup
        int jn = j + per_xcd;
        if (jn >= items) break;
        { GB_TILE(jn, mbn, nbn); (void)nbn; if (mbn >= nmb) break; }
        j = jn;
    }
#undef GB_PSTEP
#undef GB_PLOAD
#undef GB_TILE
#undef GB_SOFF
}

// ---- the default (K % 64 == 0): persistent 128 x 128 workgroups, W by LDS-DMA, X two K steps ahead -----------------------------------
// What tools/ubench/mfma_lds_bf16.hip and the round-4 ablations measured on the persistent kernel above (profiles/r04_mfma_lds_bf16.jsonl,
// r04_ablate_gemm.json): the step's twelve ds_write_b128 cost a fifth of the matrix-core rate (0.93 -> 0.73 of the register-only loop), the
// X loads another 12 % (issued one K step = ~1 us ahead, less than the HBM latency under load) and the W loads 4 %.  Here
//   * W never passes through registers: each wavefront issues six `buffer_load_dwordx4 ... lds` per step that copy the packed tile image of
//     the NEXT step straight into the other of two W stages (unpadded: fragment reads of consecutive 16-byte units are conflict-free);
//   * X keeps its register staging (it has to be split) but two register sets alternate, so a step's loads are issued two steps ahead;
//   * LDS = X terms (24.2 KB bf16x3, 16.1 KB f16x2) + 2 x 24 KB W: two workgroups per CU.
// The K loop is unrolled by two (register set / W stage = parity of the step): K % 64 == 0, other K run the kernel above.  Arithmetic per
// output element unchanged.
#define GD_WSTAGE(F16) GB_W_TILE_UNITS(F16)   // units per W stage (one packed tile image)
#define GD_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))      /* vmcnt(n) only */
// Measurement build only (tools/gemm_timeline.py compiles THIS file a second time with -DGD_PROF into tools/ubench/libgemm_prof.so; the product
// library never defines it): s_memtime stamps of the four wavefronts of one mid-grid workgroup over its first 12 K steps.
#ifdef GD_PROF
__device__ unsigned long long gd_prof[4][64];
#define GD_STAMP(k) do { if (blockIdx.x == (gridDim.x / 2 | 3u) && lane == 0 && pstep < 12) { __builtin_amdgcn_sched_barrier(0); gd_prof[wid][5 * pstep + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define GD_STAMP(k) do { } while (0)
#endif
// FLAGS & 8 (round 6): the X rows are the tokens of LoFTR's fine-level 5x5 WINDOWS, read straight from the NHWC fine map (row g = window g / W^2,
// token g % W^2 -> pixel (cell centre - W / 2 + token offset) of image img_ids[window]; outside the map: the zero row = F.unfold's padding), and the
// epilogue adds one bias row PER WINDOW (the coarse half of merge_feat, constant over a window's tokens).  Replaces fine_gather_kernel + the
// [2 M, 25, 128] window tensor + the broadcast add behind the GEMM (upstream FinePreprocess.forward: unfold -> gather at the matches -> merge_feat).
struct GdWindows {
    const int *img_ids, *cell_ids;       // [nwin]
    const float *zero;                   // K zeros
    const float *rgb;                    // [nwin, N] per-window bias, or NULL
    int wc, stride, Hf, Wf, win;
};

template <int FLAGS, bool F16>
__global__ void __launch_bounds__(256, 2) gemm_split_d_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, unsigned wp_bytes, const float *__restrict__ oscale,
                                                              const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K, int nnb, int nmb,
                                                              long long xs, long long ws, long long ys, int *guard,
                                                              const float *__restrict__ ln_gamma, const float *__restrict__ ln_beta, float ln_eps, GdWindows wnd)
{
    GB_BATCH_OFFSETS();
    constexpr int XT = GB_XT(F16);
    constexpr int WT = GB_WT(F16), WSTAGE = GD_WSTAGE(F16);
    __shared__ uint4 lds[XT * GB_TERM_UNITS + 2 * WSTAGE];
    // FLAGS & 4 (round 6): LayerNorm over the N = 128 output features in the epilogue (LoFTR's fine-level encoder layers: norm1 behind `merge`,
    // norm2 (+ residual, FLAGS & 2: Y += ...) behind the MLP's second layer -- upstream LoFTREncoderLayer.forward).  At the fine level every one of
    // these kernels is HBM-bound (2.4 M rows x 128 floats per tensor), so the separate LayerNorm pass cost its full read + write of the tensor;
    // the ~700 extra instructions per tile of the in-register reduction below are free.  Row statistics: the thread's two columns, a butterfly
    // over the 32 lanes that hold the row's other columns of this wavefront, the two wavefronts of a row exchanged through 2 KB of LDS (summed
    // in the fixed order wn = 0, 1); mean first, then the centred squares (the two-pass form of layernorm_kernel, loftr_fused.hip).
    __shared__ float lnx[(FLAGS & 4) ? 512 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nkb = K / GB_BK;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int items = ((nmb + 7) >> 3) * nnb;
#define GD_TILE(j, mb_, nb_) const int nb_ = (j) % nnb, mb_ = ((j) / nnb) * 8 + xcd
    int j = slot;
    if (j >= items) return;
    { GD_TILE(j, mb, nb); (void)nb; if (mb >= nmb) return; }
    // the item after jj in this workgroup's walk, or jj itself at the end (the streams then re-read valid memory that is never used)
    auto next_item = [&](int jj) { int jn = jj + per_xcd; if (jn < items) { GD_TILE(jn, mbn, nbn); (void)nbn; if (mbn >= nmb) jn = items; } return jn < items ? jn : jj; };

    int xdst[2], xr[2], xk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 256 * i;
        xr[i] = u >> 2; xk[i] = 8 * (u & 3);
        xdst[i] = (u & 3) * GB_KG_STRIDE + xr[i];
    }
    const int arow = (lane >> 5) * GB_KG_STRIDE + 64 * wm + (lane & 31);
    const int brow = XT * GB_TERM_UNITS + (lane >> 5) * 128 + 64 * wn + (lane & 31);
    const unsigned rowb = (unsigned)ldy * 4u;
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, bias ? N * 4 : 0, GB_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rscale = __builtin_amdgcn_make_buffer_rsrc((void *)oscale, 0, F16 ? nnb * GB_BN * 4 : 0, GB_RSRC_FLAGS);

    // W stream (LDS-DMA, one step ahead): descriptor over the whole packed weight, scalar offset = tile image + this wavefront's chunks
    typedef unsigned gd_u32x4 __attribute__((ext_vector_type(4)));
    gd_u32x4 wdesc;
    wdesc.x = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)Wp);
    wdesc.y = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)Wp >> 32) & 0xffffu);
    wdesc.z = wp_bytes;
    wdesc.w = GB_RSRC_FLAGS;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) uint4 *)lds;
    const unsigned lane16 = 16u * (unsigned)lane;
    int wj = j, wk = 0;                                    // item / K step the W stream is at
    auto wdma = [&](int stage) {
        GD_TILE(wj, mbw, nbw); (void)mbw;
        const unsigned img = (unsigned)(nbw * nkb + wk) * (unsigned)(WSTAGE * 16);
#pragma unroll
        for (int q = 0; q < 2 * WT; ++q) {
            const unsigned so = __builtin_amdgcn_readfirstlane(img + (unsigned)(64 * (wid + 4 * q)) * 16u);
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + 16u * (unsigned)(XT * GB_TERM_UNITS + stage * WSTAGE + 64 * (wid + 4 * q)));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(lane16), "s"(wdesc), "s"(so) : "memory");
        }
        if (++wk == nkb) { wk = 0; wj = next_item(wj); }
    };
    // X stream (registers, two steps ahead)
    const float *lx0, *lx1;
    int lk = 0, lj = j;
    auto window_row = [&](int g) -> const float * {
        const int ww = wnd.win * wnd.win, m = g / ww, t = g - m * ww, ky = t / wnd.win;
        const int img = wnd.img_ids[m], cell = wnd.cell_ids[m];
        const int y = (cell / wnd.wc) * wnd.stride - wnd.win / 2 + ky, x = (cell % wnd.wc) * wnd.stride - wnd.win / 2 + (t - ky * wnd.win);
        return (y >= 0 && y < wnd.Hf && x >= 0 && x < wnd.Wf) ? X + (((size_t)img * wnd.Hf + y) * wnd.Wf + x) * (size_t)ldx : wnd.zero;
    };
    auto x_tile_start = [&](int jj) {
        GD_TILE(jj, mb, nb); (void)nb;
        if constexpr ((FLAGS & 8) != 0) {
            lx0 = window_row(min(mb * GB_BM + xr[0], M - 1)) + xk[0];
            lx1 = window_row(min(mb * GB_BM + xr[1], M - 1)) + xk[1];
        } else {
            lx0 = X + (size_t)min(mb * GB_BM + xr[0], M - 1) * ldx + xk[0];
            lx1 = X + (size_t)min(mb * GB_BM + xr[1], M - 1) * ldx + xk[1];
        }
    };
    float4 xa0, xa1, xb0, xb1, xc0, xc1, xd0, xd1;         // set 0: xa (rows u >> 2), xb (+ 64 rows); set 1: xc, xd
#define GD_XLOAD(p0, p1, q0, q1) do { \
        p0 = *(const float4 *)lx0; p1 = *(const float4 *)(lx0 + 4); q0 = *(const float4 *)lx1; q1 = *(const float4 *)(lx1 + 4); \
        lx0 += GB_BK; lx1 += GB_BK; \
        if (++lk == nkb) { lk = 0; lj = next_item(lj); x_tile_start(lj); } } while (0)

    f32x16 acc[2][2];
    unsigned ov[2][2][16];                                 // FLAGS & 2: the tile of Y, fetched during the last K step
#ifdef GD_PROF
    int pstep = 0;
#define GD_PSTEP_INC ++pstep
#else
#define GD_PSTEP_INC (void)0
#endif
#define GD_SOFF(i, r) ((unsigned)(32 * (i) + ((r) & 3) + 8 * ((r) >> 2)) * rowb)
    // step of parity P: X register set P -> the X stage; W stage P (filled by the DMA of the previous step) is multiplied; the DMA of the next
    // step's W goes to stage P ^ 1 and set P is reloaded with the X of two steps ahead.  Younger than the DMA this step waits for: the previous
    // step's 4 X loads, this step's 2 WT DMAs and 4 X loads.
#define GD_STEP(P, p0, p1, q0, q1, LAST) do { \
        GD_STAMP(0); \
        __syncthreads(); \
        GD_STAMP(1); \
        gb_xsplit_store<F16>(lds, p0, p1, xdst[0]); gb_xsplit_store<F16>(lds, q0, q1, xdst[1]); \
        wdma((P) ^ 1); GD_XLOAD(p0, p1, q0, q1); \
        GD_STAMP(2); \
        if ((LAST) && (FLAGS & 2)) { \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) \
                        ov[i][jj][r] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff + 128u * jj, GD_SOFF(i, r), 0); \
            GD_VMCNT(63); } \
        else GD_VMCNT(8 + 2 * WT); \
        GD_STAMP(3); \
        __syncthreads(); \
        GD_STAMP(4); \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
            GbFrag a[2][3], b[2][3]; \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
                _Pragma("unroll") for (int t = 0; t < XT; ++t) a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i]; \
                _Pragma("unroll") for (int t = 0; t < WT; ++t) b[i][t].q = lds[(P) * WSTAGE + t * 512 + 2 * ks * 128 + brow + 32 * i]; \
                if (F16) b[i][2].q = gb_wq(b[i][0].q); } \
            gb_products<F16>(acc, a, b); } \
        GD_PSTEP_INC; } while (0)

    x_tile_start(j);
    wdma(0);
    GD_XLOAD(xa0, xa1, xb0, xb1);
    GD_XLOAD(xc0, xc1, xd0, xd1);
    for (;;) {
        GD_TILE(j, mb, nb);
        const int m0 = mb * GB_BM;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(Y + (size_t)m0 * ldy), 0, (int)((unsigned)min(GB_BM, M - m0) * rowb), GB_RSRC_FLAGS);
        const int n0 = nb * GB_BN + 64 * wn + (lane & 31);
        const unsigned yoff = (unsigned)(64 * wm + 4 * (lane >> 5)) * rowb + 4u * (unsigned)n0;
        const float bv0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0, 0, 0));
        const float bv1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0 + 128u, 0, 0));
        const float os0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rscale, 4u * (unsigned)n0, 0, 0));
        const float os1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rscale, 4u * (unsigned)n0 + 128u, 0, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
        for (int kb = 0; kb < nkb - 2; kb += 2) { GD_STEP(0, xa0, xa1, xb0, xb1, false); GD_STEP(1, xc0, xc1, xd0, xd1, false); }
        GD_STEP(0, xa0, xa1, xb0, xb1, false);
        GD_STEP(1, xc0, xc1, xd0, xd1, true);
        if (F16 && guard) {
            // range guard (guard.h): an out-of-range x[m, k] makes EVERY accumulator of output row m non-finite, so one column per row is enough:
            // column block jj = 0 of every (i, r) covers the tile's 128 rows (lanes / the two wn wavefronts repeat them)
            float chk = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) MFR_GUARD_ACC(chk, acc[i][0][r]);
            mfr_guard_commit(guard, chk);
        }
        if constexpr ((FLAGS & 4) != 0) {
            const float g0 = ln_gamma[n0], g1 = ln_gamma[n0 + 32], be0 = ln_beta[n0], be1 = ln_beta[n0 + 32];
            float part[2][16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][0][r] = F16 ? __builtin_fmaf(acc[i][0][r], os0, bv0) : acc[i][0][r] + bv0;
                    acc[i][1][r] = F16 ? __builtin_fmaf(acc[i][1][r], os1, bv1) : acc[i][1][r] + bv1;
                    part[i][r] = acc[i][0][r] + acc[i][1][r];
                }
#pragma unroll
            for (int m = 1; m <= 16; m <<= 1)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][r] += __shfl_xor(part[i][r], m, 64);
#define GD_LNROW(i, r) (64 * wm + 32 * (i) + ((r) & 3) + 8 * ((r) >> 2) + 4 * (lane >> 5))
            if ((lane & 31) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) lnx[wn * 128 + GD_LNROW(i, r)] = part[i][r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mean = (lnx[GD_LNROW(i, r)] + lnx[128 + GD_LNROW(i, r)]) * (1.0f / 128.0f);
                    acc[i][0][r] -= mean; acc[i][1][r] -= mean;
                    part[i][r] = acc[i][0][r] * acc[i][0][r] + acc[i][1][r] * acc[i][1][r];
                }
#pragma unroll
            for (int m = 1; m <= 16; m <<= 1)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][r] += __shfl_xor(part[i][r], m, 64);
            if ((lane & 31) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) lnx[256 + wn * 128 + GD_LNROW(i, r)] = part[i][r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float rstd = rsqrtf((lnx[256 + GD_LNROW(i, r)] + lnx[384 + GD_LNROW(i, r)]) * (1.0f / 128.0f) + ln_eps);
                    float y0 = acc[i][0][r] * rstd * g0 + be0, y1 = acc[i][1][r] * rstd * g1 + be1;
                    if (FLAGS & 2) { y0 = __builtin_bit_cast(float, ov[i][0][r]) + y0; y1 = __builtin_bit_cast(float, ov[i][1][r]) + y1; }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), ry, yoff, GD_SOFF(i, r), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), ry, yoff + 128u, GD_SOFF(i, r), 0);
                }
#undef GD_LNROW
        } else {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (n0 + 32 * jj >= N) continue;
            const float bv = jj ? bv1 : bv0, os = jj ? os1 : os0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = F16 ? __builtin_fmaf(acc[i][jj][r], os, bv) : acc[i][jj][r] + bv;
                    if (FLAGS & 1) v = fmaxf(v, 0.f);
                    if (FLAGS & 2) v += __builtin_bit_cast(float, ov[i][jj][r]);
                    if constexpr ((FLAGS & 8) != 0) {
                        if (wnd.rgb) {
                            const int g = min(m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), M - 1);
                            v += wnd.rgb[(size_t)(g / (wnd.win * wnd.win)) * N + n0 + 32 * jj];
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, yoff + 128u * jj, GD_SOFF(i, r), 0);
                }
        }
        }
        const int jn = next_item(j);
        if (jn == j) break;
        j = jn;
    }
    GD_VMCNT(0);                                           // the streams' last (unused) DMA must not land in the LDS of the next workgroup
#undef GD_STEP
#undef GD_SOFF
#undef GD_XLOAD
#undef GD_TILE
}

// ---- round 6: the MLP of a fine-level LoFTR encoder layer in ONE kernel ------------------------------------------------------------------------
// Upstream LoFTREncoderLayer.forward (un-vendored; call site matchers.py:50):  message = mlp(cat[x, message]); message = norm2(message);
// return x + message  with  mlp = Linear(2 C, 2 C, no bias) -> ReLU -> Linear(2 C, C, no bias),  C = 128 at the fine level.  As separate launches the
// hidden activations (2.4 M rows x 256 floats per 16 pairs) are written and read back once, and the input is read by one kernel and the residual by the
// next: 8 row-tensor passes where the arithmetic needs 3 (input, residual = its left half, output).  Here a workgroup keeps a 128-row tile on chip:
//   for each 128-feature half h of the hidden layer:
//     phase 1  hid_h^T = W1[h] X^T            K1 / 32 steps of gemm_split_d_kernel's pipeline (X through registers two steps ahead, W by LDS-DMA one
//                                             step ahead) with the MFMA operands SWAPPED: the accumulators hold hidden FEATURES along the registers and
//                                             rows along the lanes -- the layout in which a thread owns runs of 4 consecutive features of one row;
//     scale / bias / ReLU in registers; v_permlane32_swap pairs the two half-wavefronts' runs into 8 consecutive features per lane;
//     phase 2  out += hid_h W2[:, h]^T        4 steps: the wavefronts that own the step's 32 hidden features split them (split_f16.h) into the X stage
//                                             of LDS -- exactly the image gb_xsplit_store makes of 8 consecutive K elements of a row -- and all four
//                                             wavefronts multiply it with the W2 tile the DMA stream delivered (standard operand roles);
//   epilogue: LayerNorm over the 128 outputs (+ residual) as in gemm_split_d_kernel<FLAGS & 4>.
// The W stream is the same 2 (K1 / 32 + 4) tile images for every row tile (W1 half 0, W2 first half, W1 half 1, W2 second half): 384 KB per tile from L2.
// Every output element is the same sum of the same products in the same order as the two-launch path, so the result agrees with it to the rounding of
// the MFMA's internal adder under operand exchange (tests/test_gpu_gemm_split.py holds both to the float64 bars).
template <bool F16>
__device__ __forceinline__ void gb_products_t(f32x16 (&acc)[2][2], const GbFrag (&a)[2][3], const GbFrag (&b)[2][3])
{
    // acc[fi][rj] (features x rows) += W-fragment fi (first operand) x X-fragment rj (second operand); same term order as gb_products
#define GB_T4(ta, tb) do { \
        if (F16) { acc[0][0] = SF_MFMA(b[0][tb].q, a[0][ta].q, acc[0][0]); acc[0][1] = SF_MFMA(b[0][tb].q, a[1][ta].q, acc[0][1]); \
                   acc[1][0] = SF_MFMA(b[1][tb].q, a[0][ta].q, acc[1][0]); acc[1][1] = SF_MFMA(b[1][tb].q, a[1][ta].q, acc[1][1]); } \
        else     { acc[0][0] = GB_MFMA_BF(b[0][tb].v, a[0][ta].v, acc[0][0]); acc[0][1] = GB_MFMA_BF(b[0][tb].v, a[1][ta].v, acc[0][1]); \
                   acc[1][0] = GB_MFMA_BF(b[1][tb].v, a[0][ta].v, acc[1][0]); acc[1][1] = GB_MFMA_BF(b[1][tb].v, a[1][ta].v, acc[1][1]); } } while (0)
    if (F16) { GB_T4(1, 2); GB_T4(0, 1); GB_T4(0, 0); }
    else     { GB_T4(1, 1); GB_T4(0, 2); GB_T4(2, 0); GB_T4(0, 1); GB_T4(1, 0); GB_T4(0, 0); }
#undef GB_T4
}

static __device__ __forceinline__ void gb_swap32(float &a, float &b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

template <bool F16, bool ACC>
__global__ void __launch_bounds__(256, 2) mlp_ln_kernel(const float *__restrict__ X, int ldx, int K1,
                                                        const uint4 *__restrict__ W1p, unsigned w1_bytes, const float *__restrict__ os1, const float *__restrict__ b1,
                                                        const uint4 *__restrict__ W2p, unsigned w2_bytes, const float *__restrict__ os2, const float *__restrict__ b2,
                                                        const float *__restrict__ ln_gamma, const float *__restrict__ ln_beta, float ln_eps,
                                                        float *__restrict__ Y, int ldy, int M, int nmb, int *guard)
{
    constexpr int XT = GB_XT(F16);
    constexpr int WT = GB_WT(F16), WSTAGE = GD_WSTAGE(F16);
    __shared__ uint4 lds[XT * GB_TERM_UNITS + 2 * WSTAGE];
    __shared__ float lnx[512];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nkb1 = K1 / GB_BK;                           // even (K1 % 64 == 0)
    const int half_steps = nkb1 + 4, steps = 2 * half_steps;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int items = (nmb + 7) >> 3;
    int j = slot;
    if (j >= items || j * 8 + xcd >= nmb) return;
    auto next_item = [&](int jj) { const int jn = jj + per_xcd; return (jn < items && jn * 8 + xcd < nmb) ? jn : jj; };

    int xdst[2], xr[2], xk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 256 * i;
        xr[i] = u >> 2; xk[i] = 8 * (u & 3);
        xdst[i] = (u & 3) * GB_KG_STRIDE + xr[i];
    }
    const int arow = (lane >> 5) * GB_KG_STRIDE + 64 * wm + (lane & 31);
    const int brow = XT * GB_TERM_UNITS + (lane >> 5) * 128 + 64 * wn + (lane & 31);
    const unsigned rowb = (unsigned)ldy * 4u;

    // W stream: the same `steps` tile images for every row tile
    typedef unsigned gd_u32x4 __attribute__((ext_vector_type(4)));
    gd_u32x4 wdesc1, wdesc2;
    wdesc1.x = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)W1p);
    wdesc1.y = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)W1p >> 32) & 0xffffu);
    wdesc1.z = w1_bytes; wdesc1.w = GB_RSRC_FLAGS;
    wdesc2.x = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)W2p);
    wdesc2.y = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)W2p >> 32) & 0xffffu);
    wdesc2.z = w2_bytes; wdesc2.w = GB_RSRC_FLAGS;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) uint4 *)lds;
    const unsigned lane16 = 16u * (unsigned)lane;
    int wstep = 0;
    auto wdma = [&](int stage) {
        const int half = wstep >= half_steps ? 1 : 0, r = wstep - half * half_steps;
        const bool first = r < nkb1;
        const unsigned img = (unsigned)(first ? half * nkb1 + r : 4 * half + (r - nkb1)) * (unsigned)(WSTAGE * 16);
#pragma unroll
        for (int q = 0; q < 2 * WT; ++q) {
            const unsigned so = __builtin_amdgcn_readfirstlane(img + (unsigned)(64 * (wid + 4 * q)) * 16u);
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + 16u * (unsigned)(XT * GB_TERM_UNITS + stage * WSTAGE + 64 * (wid + 4 * q)));
            if (first) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(lane16), "s"(wdesc1), "s"(so) : "memory");
            else       asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(lane16), "s"(wdesc2), "s"(so) : "memory");
        }
        if (++wstep == steps) wstep = 0;
    };
    // X stream (registers, two steps ahead): the tile's K1 / 32 steps twice (once per hidden half), then the next tile
    const float *lx0, *lx1;
    int xstep = 0, lj = j;
    auto x_tile_start = [&](int jj) {
        const int mb = jj * 8 + xcd;
        lx0 = X + (size_t)min(mb * GB_BM + xr[0], M - 1) * ldx + xk[0];
        lx1 = X + (size_t)min(mb * GB_BM + xr[1], M - 1) * ldx + xk[1];
    };
    float4 xa0, xa1, xb0, xb1, xc0, xc1, xd0, xd1;
#define ML_XLOAD(p0, p1, q0, q1) do { \
        p0 = *(const float4 *)lx0; p1 = *(const float4 *)(lx0 + 4); q0 = *(const float4 *)lx1; q1 = *(const float4 *)(lx1 + 4); \
        lx0 += GB_BK; lx1 += GB_BK; ++xstep; \
        if (xstep == nkb1) { lx0 -= K1; lx1 -= K1; } \
        else if (xstep == 2 * nkb1) { xstep = 0; lj = next_item(lj); x_tile_start(lj); } } while (0)

    f32x16 acc1[2][2], acc2[2][2];
#define ML_SOFF(i, r) ((unsigned)(32 * (i) + ((r) & 3) + 8 * ((r) >> 2)) * rowb)
    // phase-1 step of parity P (X register set P, W stage P); FIRST: no X loads of a previous phase-1 step are in flight behind this step's DMA
#define ML_STEP1(P, p0, p1, q0, q1, FIRST) do { \
        __syncthreads(); \
        gb_xsplit_store<F16>(lds, p0, p1, xdst[0]); gb_xsplit_store<F16>(lds, q0, q1, xdst[1]); \
        wdma((P) ^ 1); ML_XLOAD(p0, p1, q0, q1); \
        if (FIRST) GD_VMCNT(4 + 2 * WT); else GD_VMCNT(8 + 2 * WT); \
        __syncthreads(); \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
            GbFrag a[2][3], b[2][3]; \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
                _Pragma("unroll") for (int t = 0; t < XT; ++t) a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i]; \
                _Pragma("unroll") for (int t = 0; t < WT; ++t) b[i][t].q = lds[(P) * WSTAGE + t * 512 + 2 * ks * 128 + brow + 32 * i]; \
                if (F16) b[i][2].q = gb_wq(b[i][0].q); } \
            gb_products_t<F16>(acc1, a, b); } } while (0)

    x_tile_start(j);
    wdma(0);
    ML_XLOAD(xa0, xa1, xb0, xb1);
    ML_XLOAD(xc0, xc1, xd0, xd1);
    const float g0 = ln_gamma[64 * wn + (lane & 31)], g1 = ln_gamma[64 * wn + 32 + (lane & 31)];
    const float be0 = ln_beta[64 * wn + (lane & 31)], be1 = ln_beta[64 * wn + 32 + (lane & 31)];
    for (;;) {
        const int mb = j * 8 + xcd, m0 = mb * GB_BM;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(Y + (size_t)m0 * ldy), 0, (int)((unsigned)min(GB_BM, M - m0) * rowb), GB_RSRC_FLAGS);
        const int n0 = 64 * wn + (lane & 31);
        const unsigned yoff = (unsigned)(64 * wm + 4 * (lane >> 5)) * rowb + 4u * (unsigned)n0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][jj][r] = 0.f;
        float chk = 0.f;
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][jj][r] = 0.f;
            ML_STEP1(0, xa0, xa1, xb0, xb1, true);
            ML_STEP1(1, xc0, xc1, xd0, xd1, false);
            for (int kb = 2; kb < nkb1; kb += 2) { ML_STEP1(0, xa0, xa1, xb0, xb1, false); ML_STEP1(1, xc0, xc1, xd0, xd1, false); }
            // range guard (guard.h): an out-of-range x poisons every hidden feature of its row: one register per (feature block 0, row block)
            if (F16 && guard) { MFR_GUARD_ACC(chk, acc1[0][0][0]); MFR_GUARD_ACC(chk, acc1[0][1][0]); }
            // scale / bias / ReLU: register r of acc1[fi][rj] is hidden feature 128 h + 64 wn + 32 fi + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 128 * h + 64 * wn + 32 * fi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float bv = b1 ? b1[f] : 0.f;
                    const float os = F16 ? os1[f] : 1.0f;
#pragma unroll
                    for (int rj = 0; rj < 2; ++rj) {
                        const float v = F16 ? __builtin_fmaf(acc1[fi][rj][r], os, bv) : acc1[fi][rj][r] + bv;
                        acc1[fi][rj][r] = fmaxf(v, 0.f);
                    }
                }
            // pair the half-wavefronts' runs: afterwards a lane of half hl holds, for g' = 0, 1, the 8 consecutive features of k group g' + 2 hl in
            // registers 4 g' .. 4 g' + 3 (features 8 kg .. + 3) and 8 + 4 g' .. 8 + 4 g' + 3 (features 8 kg + 4 .. + 7)
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int rj = 0; rj < 2; ++rj)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        float a_ = acc1[fi][rj][r], b_ = acc1[fi][rj][r + 8];
                        gb_swap32(a_, b_);
                        acc1[fi][rj][r] = a_; acc1[fi][rj][r + 8] = b_;
                    }
            // phase 2: out += hid_h W2[:, 128 h + 32 c .. + 32)^T, c = 0 .. 3 (W stage = c & 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                __syncthreads();
                if (wn == (c >> 1)) {
                    const int fi = c & 1;
#pragma unroll
                    for (int rj = 0; rj < 2; ++rj)
#pragma unroll
                        for (int gq = 0; gq < 2; ++gq) {
                            const float4 p = make_float4(acc1[fi][rj][4 * gq], acc1[fi][rj][4 * gq + 1], acc1[fi][rj][4 * gq + 2], acc1[fi][rj][4 * gq + 3]);
                            const float4 q = make_float4(acc1[fi][rj][8 + 4 * gq], acc1[fi][rj][9 + 4 * gq], acc1[fi][rj][10 + 4 * gq], acc1[fi][rj][11 + 4 * gq]);
                            gb_xsplit_store<F16>(lds, p, q, (gq + 2 * (lane >> 5)) * GB_KG_STRIDE + 64 * wm + 32 * rj + (lane & 31));
                        }
                }
                wdma((c & 1) ^ 1);
                GD_VMCNT(2 * WT);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    GbFrag a[2][3], b[2][3];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
#pragma unroll
                        for (int t = 0; t < XT; ++t) a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i];
#pragma unroll
                        for (int t = 0; t < WT; ++t) b[i][t].q = lds[(c & 1) * WSTAGE + t * 512 + 2 * ks * 128 + brow + 32 * i];
                        if (F16) b[i][2].q = gb_wq(b[i][0].q);
                    }
                    gb_products<F16>(acc2, a, b);
                }
            }
        }
        // ---- epilogue: out = acc2 / scale + bias -> LayerNorm (+ residual) -> Y
        unsigned ov[2][2][16];
        if (ACC) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ov[i][jj][r] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff + 128u * jj, ML_SOFF(i, r), 0);
        }
        if (F16 && guard) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) MFR_GUARD_ACC(chk, acc2[i][0][r]);
            mfr_guard_commit(guard, chk);
        }
        {
            const float bv0 = b2 ? b2[n0] : 0.f, bv1 = b2 ? b2[n0 + 32] : 0.f;
            const float os0 = F16 ? os2[n0] : 1.f, os1v = F16 ? os2[n0 + 32] : 1.f;
            float part[2][16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc2[i][0][r] = F16 ? __builtin_fmaf(acc2[i][0][r], os0, bv0) : acc2[i][0][r] + bv0;
                    acc2[i][1][r] = F16 ? __builtin_fmaf(acc2[i][1][r], os1v, bv1) : acc2[i][1][r] + bv1;
                    part[i][r] = acc2[i][0][r] + acc2[i][1][r];
                }
#pragma unroll
            for (int m = 1; m <= 16; m <<= 1)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][r] += __shfl_xor(part[i][r], m, 64);
#define ML_LNROW(i, r) (64 * wm + 32 * (i) + ((r) & 3) + 8 * ((r) >> 2) + 4 * (lane >> 5))
            if ((lane & 31) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) lnx[wn * 128 + ML_LNROW(i, r)] = part[i][r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mean = (lnx[ML_LNROW(i, r)] + lnx[128 + ML_LNROW(i, r)]) * (1.0f / 128.0f);
                    acc2[i][0][r] -= mean; acc2[i][1][r] -= mean;
                    part[i][r] = acc2[i][0][r] * acc2[i][0][r] + acc2[i][1][r] * acc2[i][1][r];
                }
#pragma unroll
            for (int m = 1; m <= 16; m <<= 1)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][r] += __shfl_xor(part[i][r], m, 64);
            if ((lane & 31) == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) lnx[256 + wn * 128 + ML_LNROW(i, r)] = part[i][r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float rstd = rsqrtf((lnx[256 + ML_LNROW(i, r)] + lnx[384 + ML_LNROW(i, r)]) * (1.0f / 128.0f) + ln_eps);
                    float y0 = acc2[i][0][r] * rstd * g0 + be0, y1 = acc2[i][1][r] * rstd * g1 + be1;
                    if (ACC) { y0 = __builtin_bit_cast(float, ov[i][0][r]) + y0; y1 = __builtin_bit_cast(float, ov[i][1][r]) + y1; }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), ry, yoff, ML_SOFF(i, r), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), ry, yoff + 128u, ML_SOFF(i, r), 0);
                }
#undef ML_LNROW
        }
        const int jn = next_item(j);
        if (jn == j) break;
        j = jn;
    }
    GD_VMCNT(0);                                           // the streams' last (unused) DMA must not land in the LDS of the next workgroup
#undef ML_STEP1
#undef ML_SOFF
#undef ML_XLOAD
}

// ---- implicit-GEMM convolution on NCHW images (round 5): what the matcher backbones' strided / 1x1 / 7x7 convolutions run ----------------------
// Reference call site: LoFTR_matcher (etc/feature_matching_baselines/matchers.py:16-19,50) -> the un-vendored ResNet-FPN backbone: conv1 (7x7, stride 2,
// 1 -> 128), the first 3x3 convolution and the 1x1 downsample of layer2.0 / layer3.0 (stride 2), the 1x1 lateral / output convolutions of the FPN
// (SURVEY.md Appendix A.4); rounds 1-4 ran them as library convolutions / batched GEMMs (MIOpen, hipBLASLt: 16 % of the LoFTR step).
//     out[b, co, oy, ox] = act( sum_k W[co, k] X_k(b, oy, ox) + bias[co] ),       f16x2 arithmetic (split_f16.h), fp32 accumulate
// K order -- MODE 0 (Cin >= 2): k = tap * Cpad + ci, Cpad = Cin rounded up to 32, X_k = in[b, ci, s oy + dy - pad, s ox + dx - pad]  (a K step = one tap,
//            32 consecutive input channels);  MODE 1 (Cin == 1): k = tap, X_k = in[b, 0, s oy + dy - pad, s ox + dx - pad]  (a K step = 32 taps).
// Zero padding and channels / taps beyond the real ones come out of the buffer range check (offset beyond the image -> 0).
// The weight matrix [Cout, K] is packed by mfr_gemm_f16x2_pack (same tile images as the linear layers, per-output-channel scale).
// Mapping: workgroup = 128 output channels x 128 consecutive output pixels (raster order) of ONE image, 4 wavefronts as 2 x 2 of 64 x 64.  The MFMA's
// first operand is the WEIGHT fragment, so an accumulator holds 32 pixels along the lanes and channels along its registers: NCHW stores of 128
// bytes.  X staging: a thread owns (pixel, k group of 8) -- eight 4-byte loads whose lanes are consecutive pixels (coalesced along x, half the
// sectors used at stride 2), split in registers, one 16-byte LDS unit per term; W through registers.  One tile per workgroup, loads one K step ahead.
// UP (round 6): the epilogue adds the 2x bilinear up-sampling (align_corners = True, the arithmetic of upsample_ac_kernel in loftr_fused.hip) of `lo`
// [B, Cout, Hl, Wl] -- the FPN merge  layerN_outconv(x_N) + interpolate(x_{N+1}_out)  of LoFTR's backbone in ONE pass: the 1x1 convolution's output is
// not written, re-read and re-written by a separate up-sample-and-add kernel (4.9 GB of the 360x272 level's traffic per 32 images).
template <int MODE, bool RELU, bool UP>
__global__ void __launch_bounds__(256, 2) conv_igemm_f16x2_kernel(const float *__restrict__ X, const uint4 *__restrict__ Wp, const float *__restrict__ oscale,
                                                                  const float *__restrict__ bias, float *__restrict__ Y, int B, int Cin, int H, int W, int Cout,
                                                                  int Ho, int Wo, int KH, int KW, int stride, int pad, int nkb, int cblocks, int nnb, int npt, int *guard,
                                                                  const float *__restrict__ lo, int Hl, int Wl, float rh, float rw)
{
    constexpr int XT = 2, WT = 2;
    __shared__ uint4 lds[(XT + WT) * GB_TERM_UNITS];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;                  // channel half, pixel half
    // channel blocks innermost: the workgroups that share a pixel tile run back to back (its input stays in L2)
    const int nb = blockIdx.x % nnb;
    const int pt = (blockIdx.x / nnb) % npt, b = blockIdx.x / (nnb * npt);
    const int HW = H * W, HoWo = Ho * Wo;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)b * Cin * HW), 0, Cin * HW * 4, GB_RSRC_FLAGS);

    // X staging: unit u = tid + 256 i -> pixel u & 127 of the tile, k group u >> 7 (wave-uniform)
    int xdst[2], kg[2];
    int iy0[2], ix0[2];                                     // input coordinates of tap (0, 0) for this thread's pixel (the same pixel for both units)
    {
        const int p = pt * 128 + (tid & 127);
        const int oy = p / Wo, ox = p - oy * Wo;
        const bool pv = p < HoWo;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            kg[i] = __builtin_amdgcn_readfirstlane((tid >> 7) + 2 * i);
            xdst[i] = kg[i] * GB_KG_STRIDE + (tid & 127);
            iy0[i] = pv ? oy * stride - pad : -(1 << 20);   // pixels beyond the image: every tap out of range
            ix0[i] = ox * stride - pad;
        }
    }
    int wdst[2 * WT];
#pragma unroll
    for (int i = 0; i < 2 * WT; ++i) {
        const int u = tid + 256 * i, term = u / 512, kgw = (u % 512) / 128, f = u % 128;
        wdst[i] = (XT + term) * GB_TERM_UNITS + kgw * GB_KG_STRIDE + f;
    }
    const uint4 *wtile = Wp + (size_t)nb * nkb * GB_W_TILE_UNITS(true) + tid;

    float xr[2][8];
    uint4 w0, w1, w2, w3;
    auto gload = [&](int kb) {
        if (MODE == 0) {
            const int tap = kb / cblocks, c0 = (kb - tap * cblocks) * 32;
            const int dy = tap / KW, dx = tap - dy * KW;
            const int iy = iy0[0] + dy, ix = ix0[0] + dx;
            const unsigned vo = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (unsigned)(iy * W + ix) * 4u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {                // channels >= Cin (padding of the last 32-channel block): the out-of-range vector offset -> 0
                    const int ch = c0 + 8 * kg[i] + e;       // (wave-uniform; the scalar offset alone could wrap for Cpad * H * W * 4 >= 2^32)
                    xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ch < Cin ? vo : 0x80000000u, (unsigned)min(ch, Cin - 1) * (unsigned)HW * 4u, 0));
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int tap = kb * 32 + 8 * kg[i] + e;
                    const int dy = tap / KW, dx = tap - dy * KW;
                    const int iy = iy0[0] + dy, ix = ix0[0] + dx;
                    const unsigned vo = (tap < KH * KW && iy >= 0 && iy < H && ix >= 0 && ix < W) ? (unsigned)(iy * W + ix) * 4u : 0x80000000u;
                    xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, 0, 0));
                }
        }
        const uint4 *wt_ = wtile + (size_t)kb * GB_W_TILE_UNITS(true);
        w0 = wt_[0]; w1 = wt_[256]; w2 = wt_[512]; w3 = wt_[768];
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned h[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sf_split2(xr[i][2 * q], xr[i][2 * q + 1], SF_LOW_SCALE, h[q], l[q]);
            lds[0 * GB_TERM_UNITS + xdst[i]] = make_uint4(h[0], h[1], h[2], h[3]);
            lds[1 * GB_TERM_UNITS + xdst[i]] = make_uint4(l[0], l[1], l[2], l[3]);
        }
        lds[wdst[0]] = w0; lds[wdst[1]] = w1; lds[wdst[2]] = w2; lds[wdst[3]] = w3;
    };

    f32x16 acc[2][2];                                       // [channel block i][pixel block j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wrow = (lane >> 5) * GB_KG_STRIDE + 64 * wm + (lane & 31);      // weight fragment rows (channels)
    const int prow = (lane >> 5) * GB_KG_STRIDE + 64 * wn + (lane & 31);      // X fragment rows (pixels)

    gload(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();                                    // the previous step's fragment reads are done
        lstore();
        __syncthreads();
        gload(min(kb + 1, nkb - 1));                        // in flight during the MFMAs below (the last step re-reads its own tile)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 wh[2], wl[2], xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wh[i] = lds[(XT + 0) * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + wrow + 32 * i];
                wl[i] = lds[(XT + 1) * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + wrow + 32 * i];
                xh[i] = lds[0 * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + prow + 32 * i];
                xl[i] = lds[1 * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + prow + 32 * i];
            }
            // small terms first: wq xl, wl xh, wh xh; the four accumulators alternate
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint4 wq = gb_wq(wh[i]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = SF_MFMA(wq, xl[j], acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = SF_MFMA(wl[i], xh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = SF_MFMA(wh[i], xh[j], acc[i][j]);
        }
    }

    if (guard) {
        // range guard (guard.h): an out-of-range input makes the accumulators of ALL output channels of the pixels it reaches non-finite:
        // one channel (i = 0, r = 0) of each of the thread's two pixels covers the tile
        float chk = 0.f;
        MFR_GUARD_ACC(chk, acc[0][0][0]); MFR_GUARD_ACC(chk, acc[0][1][0]);
        mfr_guard_commit(guard, chk);
    }
    // epilogue: register r of tile (i, j): channel 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), pixel 64 wn + 32 j + (lane & 31).
    // Round 6 (what conv_direct.hip's store tail measured, profiles/r06_dconv_timeline_*.json): gfx950 returns loads and stores through ONE in-order
    // counter, so a load that is waited for behind a store waits for the store's acknowledgement (~1 k cycles), and a predicated plain store costs
    // a compare, two exec-mask updates and a branch.  The first form of this epilogue -- 32 x [load scale, bias (, eight up-sampling taps); wait; two
    // predicated stores] -- cost more than the whole K loop of a 1x1 / 7x7 layer.  Now: scale / bias of all 32 registers are fetched before the
    // first store, the up-sampling taps two iterations ahead of their use, and every store is an unconditional buffer store whose offset lies
    // beyond the buffer where nothing must be written (invalid channel part 0x40000000, invalid pixel part 0x80000000: host check Cout Ho Wo 4 < 2^30).
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)(Y + (size_t)b * Cout * HoWo), 0, Cout * HoWo * 4, GB_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void *)(UP ? lo + (size_t)b * Cout * ((size_t)Hl * Wl) : nullptr), 0, UP ? Cout * Hl * Wl * 4 : 0, GB_RSRC_FLAGS);
    float osv[2][16], bvv[2][16];
    unsigned cho[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = nb * GB_BN + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            osv[i][r] = oscale[co];                                              // (padded to nnb * 128 entries)
            bvv[i][r] = bias ? bias[min(co, Cout - 1)] : 0.f;
            cho[i][r] = co < Cout ? (unsigned)co : 0x10000000u;                  // in ELEMENTS of a plane (x plane size x 4 below)
        }
    unsigned pixoff[2];
    // UP: the four taps and weights of this thread's two pixels (the same for every channel)
    unsigned uo[2], udw[2], udh[2];
    float uh0[2], uh1[2], uw0[2], uw1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = pt * 128 + 64 * wn + 32 * j + (lane & 31);
        pixoff[j] = p < HoWo ? (unsigned)p * 4u : 0x80000000u;
        if (UP) {
            const int pc = min(p, HoWo - 1);
            const int oy = pc / Wo, ox = pc - oy * Wo;
            const float h1r = rh * oy, w1r = rw * ox;
            const int h1 = min((int)h1r, Hl - 1), w1 = min((int)w1r, Wl - 1);
            uh1[j] = h1r - h1; uh0[j] = 1.f - uh1[j];
            uw1[j] = w1r - w1; uw0[j] = 1.f - uw1[j];
            uo[j] = (unsigned)(h1 * Wl + w1) * 4u; udw[j] = (w1 < Wl - 1) ? 4u : 0u; udh[j] = (h1 < Hl - 1) ? (unsigned)Wl * 4u : 0u;
        }
    }
    const unsigned HlWl4 = (unsigned)(Hl * Wl) * 4u, HoWo4 = (unsigned)HoWo * 4u;
    constexpr int UD = 2;                                                        // iterations of up-sampling taps in flight ahead of the stores
    float ut[UP ? 32 : 1][2][4];
    auto uload = [&](int it) {
        if (UP) {
            const int i = it >> 4, r = it & 15;
            const unsigned cb = cho[i][r] < 0x10000000u ? cho[i][r] * HlWl4 : 0x80000000u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ut[it][j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, cb + uo[j], 0, 0));
                ut[it][j][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, cb + uo[j] + udw[j], 0, 0));
                ut[it][j][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, cb + uo[j] + udh[j], 0, 0));
                ut[it][j][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, cb + uo[j] + udh[j] + udw[j], 0, 0));
            }
        }
    };
    if (UP) {
#pragma unroll
        for (int it = 0; it < UD; ++it) uload(it);
    }
#pragma unroll
    for (int it = 0; it < 32; ++it) {
        const int i = it >> 4, r = it & 15;
        if (UP && it + UD < 32) uload(it + UD);
        const unsigned cb = cho[i][r] < 0x10000000u ? cho[i][r] * HoWo4 : 0x40000000u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v = __builtin_fmaf(acc[i][j][r], osv[i][r], bvv[i][r]);
            if (RELU) v = fmaxf(v, 0.f);
            if (UP) v += uh0[j] * (uw0[j] * ut[it][j][0] + uw1[j] * ut[it][j][1]) + uh1[j] * (uw0[j] * ut[it][j][2] + uw1[j] * ut[it][j][3]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, cb + pixoff[j], 0, 0);
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------
static size_t gb_tile_bytes(int N, int K, bool f16) { return (size_t)((N + GB_BN - 1) / GB_BN) * (K / GB_BK) * GB_W_TILE_UNITS(f16) * 16; }

template <bool F16>
static int gb_launch(const float *x, int ldx, const void *packed_w, const float *bias, float *y, int ldy, int M, int N, int K, int flags, void *stream,
                     int nbatch = 1, long long xs = 0, long long ws_bytes = 0, long long ys = 0, const float *ln_gamma = nullptr, const float *ln_beta = nullptr,
                     float ln_eps = 0.f)
{
    if (ln_gamma) {
        // LayerNorm epilogue: the default (LDS-DMA) kernel only, one 128-feature block; flags: 0 or 2 (Y += LayerNorm(...))
        if (!ln_beta || N != GB_BN || (K % 64) || (flags & ~2) || nbatch != 1 || !x || !packed_w || !y || M <= 0 || (ldx & 3) || ldx < K || ldy < N || ((uintptr_t)x & 15)) return MFR_E_ARG;
        const size_t tb = gb_tile_bytes(N, K, F16);
        if (tb >= 0xffffffffull) return MFR_E_ARG;
        const int nmb = (M + GB_BM - 1) / GB_BM;
        const long long per_xcd = (long long)((nmb + 7) / 8);
        const unsigned grid = 8u * (unsigned)(per_xcd < 64 ? per_xcd : 64);
        const float *oscale = F16 ? (const float *)((const char *)packed_w + tb) : nullptr;
        int *g = F16 ? mfr_guard_current() : (int *)nullptr;
        if (flags & 2) hipLaunchKernelGGL((gemm_split_d_kernel<6, F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, (const uint4 *)packed_w, (unsigned)tb, oscale, bias, y, ldy, M, N, K, 1, nmb, 0ll, 0ll, 0ll, g, ln_gamma, ln_beta, ln_eps, GdWindows{});
        else           hipLaunchKernelGGL((gemm_split_d_kernel<4, F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, (const uint4 *)packed_w, (unsigned)tb, oscale, bias, y, ldy, M, N, K, 1, nmb, 0ll, 0ll, 0ll, g, ln_gamma, ln_beta, ln_eps, GdWindows{});
        CHECK_LAUNCH();
        return 0;
    }
    if (nbatch <= 0 || nbatch > 65535 || (ws_bytes & 15) || (xs & 3) || xs < 0 || ys < 0) return MFR_E_ARG;
    const long long ws = ws_bytes / 16;
    // flags: 1 = ReLU, 2 = accumulate; 4 = one tile per workgroup (the baseline of the bitwise-agreement test), 8 = persistent 128 x 128 workgroups
    // with register-staged W; no variant flag: W by LDS-DMA (K % 64 == 0, packed weight < 4 GB), else as flag 8
    if (!x || !packed_w || !y || M <= 0 || N <= 0 || K <= 0 || (K % GB_BK) || (ldx & 3) || ldx < K || ldy < N || flags < 0 || (flags & ~15) || (flags & 12) == 12) return MFR_E_ARG;
    if (((uintptr_t)x & 15)) return MFR_E_ARG;
    const int f = flags & 3, one_tile = flags & 4;
    int pk = flags & 8;
    const size_t tb = gb_tile_bytes(N, K, F16);
    if (!one_tile && !pk && ((K % 64) || tb >= 0xffffffffull)) pk = 8;
    const int nnb = (N + GB_BN - 1) / GB_BN, nmb = (M + GB_BM - 1) / GB_BM;
    const long long tiles = (long long)nmb * nnb;
    if (tiles > 0x7fffffffll) return MFR_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const uint4 *wp = (const uint4 *)packed_w;
    const float *oscale = F16 ? (const float *)((const char *)packed_w + tb) : nullptr;
    // 2 workgroups per CU on 256 CUs; fewer when there are fewer tiles (multiple of 8: one share per XCD)
    const long long per_xcd = (long long)((nmb + 7) / 8) * nnb;
    // f16x2 without the accumulating epilogue: 158 registers, 48 KB of LDS -> three workgroups per CU (0.119 -> 0.099 ms on the 256 -> 768 layer,
    // profiles/r05_ab_gemm.json); the others: two
    const long long cap = (F16 && !(f & 2) && !pk) ? 96 : 64;
    const unsigned grid = 8u * (unsigned)(per_xcd < cap ? per_xcd : cap);
#define GB_SW(GO) switch (f) { case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; default: GO(3); break; }
    if (one_tile) {
#define GB_GO(F) hipLaunchKernelGGL((gemm_split_kernel<F, F16>), dim3((unsigned)tiles, (unsigned)nbatch), dim3(256), 0, st, x, ldx, wp, oscale, bias, y, ldy, M, N, K, nnb, xs, ws, ys, F16 ? mfr_guard_current() : (int *)nullptr)
        GB_SW(GB_GO)
#undef GB_GO
    } else if (pk) {
#define GB_GO(F) hipLaunchKernelGGL((gemm_split_pk_kernel<F, F16>), dim3(grid, (unsigned)nbatch), dim3(256), 0, st, x, ldx, wp, oscale, bias, y, ldy, M, N, K, nnb, nmb, xs, ws, ys, F16 ? mfr_guard_current() : (int *)nullptr)
        GB_SW(GB_GO)
#undef GB_GO
    } else {
#define GB_GO(F) hipLaunchKernelGGL((gemm_split_d_kernel<F, F16>), dim3(grid, (unsigned)nbatch), dim3(256), 0, st, x, ldx, wp, (unsigned)tb, oscale, bias, y, ldy, M, N, K, nnb, nmb, xs, ws, ys, F16 ? mfr_guard_current() : (int *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, GdWindows{})
        GB_SW(GB_GO)
#undef GB_GO
    }
#undef GB_SW
    CHECK_LAUNCH();
    return 0;
}

template <bool F16>
static int gb_mlp_ln(const float *x, int ldx, int K1, const void *packed_w1, const float *b1, const void *packed_w2, const float *b2, const float *gamma, const float *beta,
                     float eps, float *y, int ldy, int M, int accumulate, void *stream)
{
    // hidden width 256 (two 128-feature blocks), output width 128: the fine-level LoFTR encoder layer (d_model 128)
    if (!x || !packed_w1 || !packed_w2 || !gamma || !beta || !y || M <= 0 || K1 <= 0 || (K1 % 64) || (ldx & 3) || ldx < K1 || ldy < GB_BN || ((uintptr_t)x & 15)) return MFR_E_ARG;
    const size_t tb1 = gb_tile_bytes(2 * GB_BN, K1, F16), tb2 = gb_tile_bytes(GB_BN, 2 * GB_BN, F16);
    if (tb1 >= 0xffffffffull) return MFR_E_ARG;
    const int nmb = (M + GB_BM - 1) / GB_BM;
    const long long per_xcd = (long long)((nmb + 7) / 8);
    const unsigned grid = 8u * (unsigned)(per_xcd < 64 ? per_xcd : 64);
    const float *os1 = F16 ? (const float *)((const char *)packed_w1 + tb1) : nullptr, *os2 = F16 ? (const float *)((const char *)packed_w2 + tb2) : nullptr;
    int *g = F16 ? mfr_guard_current() : (int *)nullptr;
    if (accumulate) hipLaunchKernelGGL((mlp_ln_kernel<F16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, K1, (const uint4 *)packed_w1, (unsigned)tb1, os1, b1,
                                       (const uint4 *)packed_w2, (unsigned)tb2, os2, b2, gamma, beta, eps, y, ldy, M, nmb, g);
    else            hipLaunchKernelGGL((mlp_ln_kernel<F16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, K1, (const uint4 *)packed_w1, (unsigned)tb1, os1, b1,
                                       (const uint4 *)packed_w2, (unsigned)tb2, os2, b2, gamma, beta, eps, y, ldy, M, nmb, g);
    CHECK_LAUNCH();
    return 0;
}

template <bool F16>
static int gb_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids, const int32_t *cell_ids, int nwin, int wc, int stride, int win,
                      const float *zero_row, const void *packed_w, const float *bias, const float *window_bias, float *y, int ldy, int N, void *stream)
{
    if (!feat || !img_ids || !cell_ids || !zero_row || !packed_w || !y || Bimg <= 0 || Hf <= 0 || Wf <= 0 || C <= 0 || (C % 64) || nwin < 0 || wc <= 0 || stride <= 0 ||
        win <= 0 || N <= 0 || ldy < N || (((uintptr_t)feat | (uintptr_t)zero_row) & 15)) return MFR_E_ARG;
    if (nwin == 0) return 0;
    const long long Mll = (long long)nwin * win * win;
    if (Mll > 0x7fffffffll) return MFR_E_ARG;
    const int M = (int)Mll, K = C;
    const size_t tb = gb_tile_bytes(N, K, F16);
    if (tb >= 0xffffffffull) return MFR_E_ARG;
    const int nnb = (N + GB_BN - 1) / GB_BN, nmb = (M + GB_BM - 1) / GB_BM;
    const long long per_xcd = (long long)((nmb + 7) / 8) * nnb;
    const unsigned grid = 8u * (unsigned)(per_xcd < 64 ? per_xcd : 64);
    const float *oscale = F16 ? (const float *)((const char *)packed_w + tb) : nullptr;
    GdWindows w{img_ids, cell_ids, zero_row, window_bias, wc, stride, Hf, Wf, win};
    hipLaunchKernelGGL((gemm_split_d_kernel<8, F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, feat, C, (const uint4 *)packed_w, (unsigned)tb, oscale, bias, y, ldy, M, N, K,
                       nnb, nmb, 0ll, 0ll, 0ll, F16 ? mfr_guard_current() : (int *)nullptr, (const float *)nullptr, (const float *)nullptr, 0.f, w);
    CHECK_LAUNCH();
    return 0;
}

extern "C" {

#ifdef GD_PROF
int mfr_gemm_split_profile(unsigned long long *out_host)      /* the stamps of the last default-kernel launch: 4 wavefronts x 64 */
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(gd_prof), sizeof(unsigned long long) * 256, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : MFR_E_LAUNCH;
}
#endif

size_t mfr_gemm_bf16x3_pack_bytes(int N, int K)
{
    if (N <= 0 || K <= 0 || (K % GB_BK)) return 0;
    return gb_tile_bytes(N, K, false);
}

size_t mfr_gemm_f16x2_pack_bytes(int N, int K)
{
    if (N <= 0 || K <= 0 || (K % GB_BK)) return 0;
    return gb_tile_bytes(N, K, true) + (size_t)((N + GB_BN - 1) / GB_BN) * GB_BN * 4;      // + the per-feature 1 / scale
}

int mfr_gemm_bf16x3_pack(const float *w, int N, int K, void *packed, void *stream)
{
    if (!w || !packed || N <= 0 || K <= 0 || (K % GB_BK)) return MFR_E_ARG;
    const long long total = (long long)(gb_tile_bytes(N, K, false) / 16);
    hipLaunchKernelGGL((gb_pack_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K, total, (const float *)nullptr, (uint4 *)packed,
                       1.0f, K, 0ll, 0ll);
    CHECK_LAUNCH();
    return 0;
}

int mfr_gemm_f16x2_pack_batched(const float *w, int ldw, int nbatch, long long w_batch_stride, int N, int K, float out_mul, void *packed, void *stream)
{
    if (ldw < K || !w || !packed || N <= 0 || K <= 0 || (K % GB_BK) || nbatch <= 0 || nbatch > 65535 || !(out_mul > 0.f)) return MFR_E_ARG;
    const size_t tb = gb_tile_bytes(N, K, true);
    const long long total = (long long)(tb / 16);
    const int npad = (N + GB_BN - 1) / GB_BN * GB_BN;
    const size_t pbytes = tb + (size_t)npad * 4;                       // one problem's packed operand (a multiple of 16)
    float *oscale = (float *)((char *)packed + tb);
    hipLaunchKernelGGL(gb_scale_kernel, dim3((unsigned)npad, (unsigned)nbatch), dim3(64), 0, (hipStream_t)stream, w, N, K, npad, oscale, out_mul, ldw, w_batch_stride, (long long)(pbytes / 4));
    hipLaunchKernelGGL((gb_pack_kernel<true>), dim3((unsigned)((total + 255) / 256), (unsigned)nbatch), dim3(256), 0, (hipStream_t)stream, w, N, K, total, (const float *)oscale,
                       (uint4 *)packed, out_mul, ldw, w_batch_stride, (long long)(pbytes / 16));
    CHECK_LAUNCH();
    return 0;
}

int mfr_gemm_f16x2_pack(const float *w, int N, int K, void *packed, void *stream)
{
    return mfr_gemm_f16x2_pack_batched(w, K, 1, 0, N, K, 1.0f, packed, stream);
}

int mfr_gemm_bf16x3(const float *x, int ldx, const void *packed_w, const float *bias, float *y, int ldy, int M, int N, int K, int flags, void *stream)
{
    return gb_launch<false>(x, ldx, packed_w, bias, y, ldy, M, N, K, flags, stream);
}

int mfr_gemm_f16x2(const float *x, int ldx, const void *packed_w, const float *bias, float *y, int ldy, int M, int N, int K, int flags, void *stream)
{
    return gb_launch<true>(x, ldx, packed_w, bias, y, ldy, M, N, K, flags, stream);
}

int mfr_mlp_ln_f16x2(const float *x, int ldx, int K1, const void *packed_w1, const float *b1, const void *packed_w2, const float *b2, const float *gamma, const float *beta,
                     float eps, float *y, int ldy, int M, int accumulate, void *stream)
{
    return gb_mlp_ln<true>(x, ldx, K1, packed_w1, b1, packed_w2, b2, gamma, beta, eps, y, ldy, M, accumulate, stream);
}

int mfr_mlp_ln_bf16x3(const float *x, int ldx, int K1, const void *packed_w1, const float *b1, const void *packed_w2, const float *b2, const float *gamma, const float *beta,
                      float eps, float *y, int ldy, int M, int accumulate, void *stream)
{
    return gb_mlp_ln<false>(x, ldx, K1, packed_w1, b1, packed_w2, b2, gamma, beta, eps, y, ldy, M, accumulate, stream);
}

int mfr_gemm_f16x2_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids, const int32_t *cell_ids, int nwin, int wc, int stride, int win,
                           const float *zero_row, const void *packed_w, const float *bias, const float *window_bias, float *y, int ldy, int N, void *stream)
{
    return gb_windows<true>(feat, Bimg, Hf, Wf, C, img_ids, cell_ids, nwin, wc, stride, win, zero_row, packed_w, bias, window_bias, y, ldy, N, stream);
}

int mfr_gemm_bf16x3_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids, const int32_t *cell_ids, int nwin, int wc, int stride, int win,
                            const float *zero_row, const void *packed_w, const float *bias, const float *window_bias, float *y, int ldy, int N, void *stream)
{
    return gb_windows<false>(feat, Bimg, Hf, Wf, C, img_ids, cell_ids, nwin, wc, stride, win, zero_row, packed_w, bias, window_bias, y, ldy, N, stream);
}

int mfr_gemm_f16x2_ln(const float *x, int ldx, const void *packed_w, const float *bias, const float *gamma, const float *beta, float eps, float *y, int ldy,
                      int M, int N, int K, int accumulate, void *stream)
{
    if (!gamma || !beta) return MFR_E_ARG;
    return gb_launch<true>(x, ldx, packed_w, bias, y, ldy, M, N, K, accumulate ? 2 : 0, stream, 1, 0, 0, 0, gamma, beta, eps);
}

int mfr_gemm_bf16x3_ln(const float *x, int ldx, const void *packed_w, const float *bias, const float *gamma, const float *beta, float eps, float *y, int ldy,
                       int M, int N, int K, int accumulate, void *stream)
{
    if (!gamma || !beta) return MFR_E_ARG;
    return gb_launch<false>(x, ldx, packed_w, bias, y, ldy, M, N, K, accumulate ? 2 : 0, stream, 1, 0, 0, 0, gamma, beta, eps);
}

int mfr_gemm_f16x2_batched(const float *x, int ldx, long long x_batch_stride, const void *packed_w, const float *bias, float *y, int ldy, long long y_batch_stride,
                           int nbatch, int M, int N, int K, int flags, void *stream)
{
    return gb_launch<true>(x, ldx, packed_w, bias, y, ldy, M, N, K, flags, stream, nbatch, x_batch_stride, (long long)mfr_gemm_f16x2_pack_bytes(N, K), y_batch_stride);
}

int mfr_conv_igemm_k(int Cin, int KH, int KW)
{
    if (Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    return Cin == 1 ? (KH * KW + 31) / 32 * 32 : KH * KW * ((Cin + 31) / 32 * 32);
}

static int gb_conv_igemm(const float *x, const void *packed_w, const float *bias, float *y, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                         int stride, int pad, int relu, const float *lo, int Hl, int Wl, void *stream)
{
    if (!x || !packed_w || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return MFR_E_ARG;
    if ((size_t)4 * ((Cin + 31) / 32 * 32) * H * W >= 0x7fffffffull) return MFR_E_ARG;    // one image (padded channel count) must fit a 2 GB buffer descriptor
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return MFR_E_ARG;
    if ((size_t)4 * Cout * Ho * Wo >= 0x40000000ull) return MFR_E_ARG;                   // one output image < 1 GB: the epilogue's out-of-buffer offsets must not wrap
    if (lo && (relu || Cin == 1 || Hl <= 0 || Wl <= 0 || Ho != 2 * Hl || Wo != 2 * Wl)) return MFR_E_ARG;
    const int K = mfr_conv_igemm_k(Cin, KH, KW), nkb = K / GB_BK, cblocks = (Cin + 31) / 32;
    const int nnb = (Cout + GB_BN - 1) / GB_BN, npt = (Ho * Wo + 127) / 128;
    const long long grid = (long long)B * npt * nnb;
    if (grid > 0x7fffffffll) return MFR_E_ARG;
    const size_t tb = gb_tile_bytes(Cout, K, true);
    const float *oscale = (const float *)((const char *)packed_w + tb);
    hipStream_t st = (hipStream_t)stream;
    // torch's area_pixel_compute_scale for align_corners = True: (in - 1) / (out - 1) in f32, 0 for a single output
    const float rh = (lo && Ho > 1) ? (float)(Hl - 1) / (float)(Ho - 1) : 0.f, rw = (lo && Wo > 1) ? (float)(Wl - 1) / (float)(Wo - 1) : 0.f;
#define GB_GO(M_, R_, U_) hipLaunchKernelGGL((conv_igemm_f16x2_kernel<M_, R_, U_>), dim3((unsigned)grid), dim3(256), 0, st, x, (const uint4 *)packed_w, oscale, bias, y, B, Cin, H, W, Cout, Ho, Wo, KH, KW, stride, pad, nkb, cblocks, nnb, npt, mfr_guard_current(), lo, Hl, Wl, rh, rw)
    if (lo) GB_GO(0, false, true);
    else if (Cin == 1) { if (relu) GB_GO(1, true, false); else GB_GO(1, false, false); }
    else               { if (relu) GB_GO(0, true, false); else GB_GO(0, false, false); }
#undef GB_GO
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv_igemm_f16x2(const float *x, const void *packed_w, const float *bias, float *y, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                         int stride, int pad, int relu, void *stream)
{
    return gb_conv_igemm(x, packed_w, bias, y, B, Cin, H, W, Cout, KH, KW, stride, pad, relu, nullptr, 0, 0, stream);
}

int mfr_conv_igemm_f16x2_upadd(const float *x, const void *packed_w, const float *bias, const float *lo, int Hl, int Wl, float *y, int B, int Cin, int H, int W,
                               int Cout, int KH, int KW, int stride, int pad, void *stream)
{
    if (!lo) return MFR_E_ARG;
    return gb_conv_igemm(x, packed_w, bias, y, B, Cin, H, W, Cout, KH, KW, stride, pad, 0, lo, Hl, Wl, stream);
}

}  // extern "C"
