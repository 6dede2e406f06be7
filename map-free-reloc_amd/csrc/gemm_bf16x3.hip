// gemm_bf16x3.hip -- Y[M, N] (+)= act(X[M, K] W[N, K]^T + bias): the linear layers of the SuperGlue / LoFTR transformers (fp32 in,
// fp32 out) on the gfx950 BF16 matrix cores at fp32 accuracy ("bf16x3").
//
// Reference call site: SuperGlue_matcher / LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-120) -> the un-vendored
// networks' Conv1d(k=1) / Linear layers; rounds 1-2 ran them as library (hipBLASLt) fp32 GEMMs.
//
// Arithmetic: every fp32 operand is split EXACTLY into three bf16 terms x = h + m + l (truncation, 8 + 8 + 8 significand bits); a
// product is the six partial products hh + hm + mh + hl + lh + mm (each exact in fp32) accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Error against an fp64 product = that of the exact-fp32 matrix instruction (tools/ubench/
// bf16x3_probe.hip, profiles/r03_bf16x3_probe.jsonl: rms 2.4e-8 vs 2.8e-8 of sum|x||w| at K = 64 .. 2304).
//
// Mapping (all kernels of this file): workgroup tile = 128 rows x 128 output features, 4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles
// (64 accumulator registers); K in steps of 32.  W is split and packed ONCE per weight set in the exact image a workgroup stages (three
// terms x [k group of 8][feature][8 bf16]); X is read as fp32 (coalesced 128-byte row pieces), split by the staging threads -- each element
// once per workgroup -- and written to LDS in the same fragment order, so every MFMA operand is one conflict-free ds_read_b128.  Output
// features run along the lanes: 128-byte stores.
//
// Kernels, newest last; every one sums each output element in the same order (bitwise-equal results, tests/test_gpu_gemm_bf16x3.py):
//   gemm_bf16x3_kernel      one tile per workgroup, register-staged prefetch, 3 workgroups per CU (rounds 1-3; flag 4)
//   gemm_bf16x3_pk_kernel   persistent workgroups walking XCD-local tile lists (flag 8; 16 = with deferred tile stores); runs K % 64 != 0
//   gemm_bf16x3_w8_kernel   256 x 128 tiles, eight wavefronts, two LDS stages, opposite-phase wavefronts (flag 32; measured no faster)
//   gemm_bf16x3_d_kernel    persistent, W by LDS-DMA into two W stages, X two K steps ahead -- the default (K % 64 == 0)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GB_BM 128
#define GB_BN 128
#define GB_BK 32
#define GB_KG_STRIDE 129                  // 16-byte units per k group (128 rows + 1 pad: conflict-free stores)
#define GB_TERM_UNITS (4 * GB_KG_STRIDE)  // units per term image
#define GB_W_TILE_UNITS 1536              // packed W tile: 3 terms x 4 k groups x 128 features, 16 bytes each

union GbFrag { bf16x8 v; unsigned u[4]; uint4 q; };

__device__ __forceinline__ void gb_split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned gb_pack(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// W [N, K] f32 row-major -> packed [n block][k block][term][k group][feature (128)][8 bf16]; one thread per 16-byte unit
__global__ void __launch_bounds__(256) gb_pack_kernel(const float *__restrict__ w, int N, int K, long long total, uint4 *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int u = (int)(t % GB_W_TILE_UNITS);
    const long long tile = t / GB_W_TILE_UNITS;
    const int nkb = K / GB_BK;
    const int kb = (int)(tile % nkb), nb = (int)(tile / nkb);
    const int term = u / 512, kg = (u % 512) / 128, f = u % 128;
    const int n = nb * GB_BN + f, k0 = kb * GB_BK + 8 * kg;
    unsigned word[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (n < N) ? w[(size_t)n * K + k0 + e] : 0.f;
        unsigned h, m, l;
        gb_split3(x, h, m, l);
        word[e] = term == 0 ? h : term == 1 ? m : l;
    }
    out[t] = make_uint4(gb_pack(word[0], word[1]), gb_pack(word[2], word[3]), gb_pack(word[4], word[5]), gb_pack(word[6], word[7]));
}

#define GB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// FLAGS: 1 = ReLU, 2 = accumulate into Y (Y += ...)
template <int FLAGS>
__global__ void __launch_bounds__(256, 2) gemm_bf16x3_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, const float *__restrict__ bias,
                                                             float *__restrict__ Y, int ldy, int M, int N, int K, int nnb)
{
    __shared__ uint4 lds[6 * GB_TERM_UNITS];             // X terms 0..2, W terms 0..2 (49.5 KB)
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
    int wdst[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = tid + 256 * i, term = u / 512, kg = (u % 512) / 128, f = u % 128;
        wdst[i] = (3 + term) * GB_TERM_UNITS + kg * GB_KG_STRIDE + f;
    }
    const uint4 *wtile = Wp + (size_t)nb * nkb * GB_W_TILE_UNITS + tid;

    // (named registers, not arrays: hipcc keeps a lambda-captured array that is written under a condition in scratch memory)
    float4 xa0, xa1, xb0, xb1;
    uint4 w0, w1, w2, w3, w4, w5;
#define GB_GLOAD(kb) do { \
        xa0 = *(const float4 *)(xrow[0] + (kb) * GB_BK); xa1 = *(const float4 *)(xrow[0] + (kb) * GB_BK + 4); \
        xb0 = *(const float4 *)(xrow[1] + (kb) * GB_BK); xb1 = *(const float4 *)(xrow[1] + (kb) * GB_BK + 4); \
        const uint4 *wt_ = wtile + (size_t)(kb) * GB_W_TILE_UNITS; \
        w0 = wt_[0]; w1 = wt_[256]; w2 = wt_[512]; w3 = wt_[768]; w4 = wt_[1024]; w5 = wt_[1280]; } while (0)
    auto xsplit_store = [&](const float4 &p, const float4 &q, int dst) {
        const float x[8] = { p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w };
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gb_split3(x[e], h[e], m[e], l[e]);
        lds[0 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(h[0], h[1]), gb_pack(h[2], h[3]), gb_pack(h[4], h[5]), gb_pack(h[6], h[7]));
        lds[1 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(m[0], m[1]), gb_pack(m[2], m[3]), gb_pack(m[4], m[5]), gb_pack(m[6], m[7]));
        lds[2 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(l[0], l[1]), gb_pack(l[2], l[3]), gb_pack(l[4], l[5]), gb_pack(l[6], l[7]));
    };
#define GB_LSTORE() do { xsplit_store(xa0, xa1, xdst[0]); xsplit_store(xb0, xb1, xdst[1]); \
        lds[wdst[0]] = w0; lds[wdst[1]] = w1; lds[wdst[2]] = w2; lds[wdst[3]] = w3; lds[wdst[4]] = w4; lds[wdst[5]] = w5; } while (0)

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
        GB_LSTORE();
        __syncthreads();
        { const int kn = min(kb + 1, nkb - 1); GB_GLOAD(kn); }   // in flight during the MFMAs below (the last step re-reads its own tile)
        // without this fence hipcc sinks the loads BELOW the 48 MFMAs (ten live 16-byte registers fewer across them) and every K step
        // pays the full memory latency before its split: load -> wait -> split -> store -> MFMA, nothing overlapped inside a workgroup
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GbFrag a[2][3], b[2][3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i];
                    b[i][t].q = lds[(3 + t) * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + brow + 32 * i];
                }
            // six partial products, small terms first; the four accumulators alternate
#define GB_PROD(ta, tb) do { \
                acc[0][0] = GB_MFMA(a[0][ta].v, b[0][tb].v, acc[0][0]); acc[0][1] = GB_MFMA(a[0][ta].v, b[1][tb].v, acc[0][1]); \
                acc[1][0] = GB_MFMA(a[1][ta].v, b[0][tb].v, acc[1][0]); acc[1][1] = GB_MFMA(a[1][ta].v, b[1][tb].v, acc[1][1]); } while (0)
            GB_PROD(1, 1); GB_PROD(0, 2); GB_PROD(2, 0); GB_PROD(0, 1); GB_PROD(1, 0); GB_PROD(0, 0);
#undef GB_PROD
        }
    }

    // epilogue: accumulator register r of tile (i, j): token row 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), feature 64 wn + 32 j + (lane & 31)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nb * GB_BN + 64 * wn + 32 * j + (lane & 31);
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                float *yp = Y + (size_t)m * ldy + n;
                float v = acc[i][j][r] + bv;
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
// DEFER: the finished tile (bias / ReLU / residual applied) moves to 64 spare registers and is stored in four groups during the first four
// K steps of the NEXT tile, each group right after that step's loads were issued.  gfx950 counts loads and stores in ONE in-order counter
// (vmcnt), so a wavefront that stores a tile and then waits for any load waits for the whole tile to reach L2 first; issued this way every
// wait for a K step's loads only covers stores that are a full K step old.
// Same arithmetic per output element as the kernel above, bit for bit (tests/test_gpu_gemm_bf16x3.py).  Grid = 2 workgroups per CU (512: the
// SuperGlue shapes are 1024 / 2048 / 3072 tiles); tiles are dealt per XCD so that the workgroups sharing an X row block share an L2 (row
// block mb lives on XCD mb % 8).  ABL (measurement only, tools/ablate_gemm.py): 1 = no output stores, 2 = no global loads after the first
// K step, 3 = no X loads after the first K step (W still streamed).
#define GB_RSRC_FLAGS 0x00020000
template <int FLAGS, int DEFER, int ABL>
__global__ void __launch_bounds__(256, 2) gemm_bf16x3_pk_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, const float *__restrict__ bias,
                                                                float *__restrict__ Y, int ldy, int M, int N, int K, int nnb, int nmb)
{
    __shared__ uint4 lds[6 * GB_TERM_UNITS];
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
    int wdst[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int u = tid + 256 * i, term = u / 512, kg = (u % 512) / 128, f = u % 128;
        wdst[i] = (3 + term) * GB_TERM_UNITS + kg * GB_KG_STRIDE + f;
    }
    const int arow = (lane >> 5) * GB_KG_STRIDE + 64 * wm + (lane & 31);
    const int brow = (lane >> 5) * GB_KG_STRIDE + 64 * wn + (lane & 31);
    const unsigned rowb = (unsigned)ldy * 4u;              // bytes per row of Y
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, bias ? N * 4 : 0, GB_RSRC_FLAGS);   // no bias: every read returns 0

    // the LOAD stream runs one K step ahead of the multiply, across tile boundaries
    const float *lx0, *lx1;
    const uint4 *lw;
    int lk, lj = j;                                        // K step / item the load stream is at
    auto load_tile_start = [&](int jj) {
        GB_TILE(jj, mb, nb);
        lx0 = X + (size_t)min(mb * GB_BM + xr[0], M - 1) * ldx + xk[0];      // rows beyond M: a valid row is read, its results are never stored
        lx1 = X + (size_t)min(mb * GB_BM + xr[1], M - 1) * ldx + xk[1];
        lw = Wp + (size_t)nb * nkb * GB_W_TILE_UNITS + tid;
        lk = 0;
    };
    float4 xa0, xa1, xb0, xb1;
    uint4 w0, w1, w2, w3, w4, w5;
#define GB_PLOAD() do { \
        if (ABL != 2 || first) { \
        if (ABL != 3 || first) { xa0 = *(const float4 *)lx0; xa1 = *(const float4 *)(lx0 + 4); xb0 = *(const float4 *)lx1; xb1 = *(const float4 *)(lx1 + 4); } \
        w0 = lw[0]; w1 = lw[256]; w2 = lw[512]; w3 = lw[768]; w4 = lw[1024]; w5 = lw[1280]; } \
        lx0 += GB_BK; lx1 += GB_BK; lw += GB_W_TILE_UNITS; \
        if (++lk == nkb) { int jn = lj + per_xcd; if (jn < items) { GB_TILE(jn, mbn_, nbn_); (void)nbn_; if (mbn_ >= nmb) jn = items; } \
                           if (jn < items) lj = jn; load_tile_start(lj); } } while (0)        /* no next tile: the last load re-reads this tile's start */
    auto xsplit_store = [&](const float4 &p, const float4 &q, int dst) {
        const float x[8] = { p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w };
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gb_split3(x[e], h[e], m[e], l[e]);
        lds[0 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(h[0], h[1]), gb_pack(h[2], h[3]), gb_pack(h[4], h[5]), gb_pack(h[6], h[7]));
        lds[1 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(m[0], m[1]), gb_pack(m[2], m[3]), gb_pack(m[4], m[5]), gb_pack(m[6], m[7]));
        lds[2 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(l[0], l[1]), gb_pack(l[2], l[3]), gb_pack(l[4], l[5]), gb_pack(l[6], l[7]));
    };

    f32x16 acc[2][2];
    // ov: the finished tile on its way out (DEFER), and -- FLAGS & 2 -- the tile of Y fetched during the last K step
    unsigned ov[2][2][16];
    __amdgpu_buffer_rsrc_t rp = rbias;                     // the buffer / lane offset / lane's first feature of the tile held in ov
    unsigned offp = 0;
    int np = 0;
    bool pending = false;
    const bool defer = DEFER && nkb >= 5;                  // four store groups + the Y fetch of the last step need five K steps
    // accumulator register r of tile (i, jj) of a wavefront: row 32 i + (r & 3) + 8 (r >> 2) (+ 64 wm + 4 (lane >> 5): the lane offset), feature + 32 jj
#define GB_SOFF(i, r) ((unsigned)(32 * (i) + ((r) & 3) + 8 * ((r) >> 2)) * rowb)
#define GB_STORE_GROUP(g) do { \
        if (np + 32 * ((g) >> 1) < N) { \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { \
                if (ABL == 1 && ov[(g) & 1][(g) >> 1][r] != 0x12345678u) continue; \
                __builtin_amdgcn_raw_buffer_store_b32(ov[(g) & 1][(g) >> 1][r], rp, offp + 128u * ((g) >> 1), GB_SOFF((g) & 1, r), 0); } } } while (0)

    load_tile_start(j);
    bool first = true;                                     // (ablations 2 / 3: only the very first K step loads)
    GB_PLOAD();
    first = false;
    for (;;) {
        GB_TILE(j, mb, nb);
        const int m0 = mb * GB_BM;
        // this tile's rows of Y as one buffer (rows beyond M fall outside it: reads return 0, stores are dropped)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(Y + (size_t)m0 * ldy), 0, (int)((unsigned)min(GB_BM, M - m0) * rowb), GB_RSRC_FLAGS);
        const int n0 = nb * GB_BN + 64 * wn + (lane & 31);
        const unsigned yoff = (unsigned)(64 * wm + 4 * (lane >> 5)) * rowb + 4u * (unsigned)n0;
        const float bv0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0, 0, 0));
        const float bv1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0 + 128u, 0, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
#define GB_PPROD(ta, tb) do { \
                acc[0][0] = GB_MFMA(a[0][ta].v, b[0][tb].v, acc[0][0]); acc[0][1] = GB_MFMA(a[0][ta].v, b[1][tb].v, acc[0][1]); \
                acc[1][0] = GB_MFMA(a[1][ta].v, b[0][tb].v, acc[1][0]); acc[1][1] = GB_MFMA(a[1][ta].v, b[1][tb].v, acc[1][1]); } while (0)
#define GB_PSTEP(LAST) do { \
            __syncthreads(); \
            xsplit_store(xa0, xa1, xdst[0]); xsplit_store(xb0, xb1, xdst[1]); \
            lds[wdst[0]] = w0; lds[wdst[1]] = w1; lds[wdst[2]] = w2; lds[wdst[3]] = w3; lds[wdst[4]] = w4; lds[wdst[5]] = w5; \
            __syncthreads(); \
            GB_PLOAD(); \
            if (!(LAST) && pending) { \
                if (kb == 0) GB_STORE_GROUP(0); else if (kb == 1) GB_STORE_GROUP(1); else if (kb == 2) GB_STORE_GROUP(2); else if (kb == 3) GB_STORE_GROUP(3); } \
            if ((LAST) && (FLAGS & 2)) { \
                _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) \
                            ov[i][jj][r] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff + 128u * jj, GB_SOFF(i, r), 0); } \
            __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
                GbFrag a[2][3], b[2][3]; \
                _Pragma("unroll") for (int t = 0; t < 3; ++t) \
                    _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
                        a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i]; \
                        b[i][t].q = lds[(3 + t) * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + brow + 32 * i]; } \
                GB_PPROD(1, 1); GB_PPROD(0, 2); GB_PPROD(2, 0); GB_PPROD(0, 1); GB_PPROD(1, 0); GB_PPROD(0, 0); } } while (0)
        for (int kb = 0; kb < nkb - 1; ++kb) GB_PSTEP(false);
        pending = false;                                   // defer: the four groups went out at K steps 0 .. 3 (nkb >= 5)
        { const int kb = nkb - 1; (void)kb; GB_PSTEP(true); }

        // the finished tile -> ov
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const float bv = jj ? bv1 : bv0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][jj][r] + bv;
                    if (FLAGS & 1) v = fmaxf(v, 0.f);
                    if (FLAGS & 2) v += __builtin_bit_cast(float, ov[i][jj][r]);
                    ov[i][jj][r] = __builtin_bit_cast(unsigned, v);
                }
        }
        rp = ry; offp = yoff; np = n0;
        // next item of this workgroup
        int jn = j + per_xcd;
        if (jn < items) { GB_TILE(jn, mbn, nbn); (void)nbn; if (mbn >= nmb) jn = items; }
        if (!defer || jn >= items) { GB_STORE_GROUP(0); GB_STORE_GROUP(1); GB_STORE_GROUP(2); GB_STORE_GROUP(3); }
        else pending = true;
        if (jn >= items) break;
        j = jn;
    }
#undef GB_PSTEP
#undef GB_PPROD
#undef GB_PLOAD
#undef GB_TILE
#undef GB_SOFF
#undef GB_STORE_GROUP
}

// ---- round 4, the default (K % 64 == 0): persistent 128 x 128 workgroups, W by LDS-DMA, X two K steps ahead -----------------------------------
// What tools/ubench/mfma_lds_bf16.hip and tools/ablate_gemm.py measured on the persistent kernel above (profiles/r04_mfma_lds_bf16.jsonl,
// r04_ablate_gemm.json): the step's twelve ds_write_b128 cost a fifth of the matrix-core rate (0.93 -> 0.73 of the register-only loop), the
// X loads another 12 % (issued one K step = ~1 us ahead, less than the HBM latency under load) and the W loads 4 %.  Here
//   * W never passes through registers: each wavefront issues six `buffer_load_dwordx4 ... lds` per step that copy the packed tile image of
//     the NEXT step straight into the other of two W stages (unpadded: fragment reads of consecutive 16-byte units are conflict-free);
//   * X keeps its register staging (it has to be split) but two register sets alternate, so a step's loads are issued two steps ahead;
//   * LDS = 24.2 KB X terms + 2 x 24 KB W = 72.2 KB: two workgroups per CU.
// The K loop is unrolled by two (register set / W stage = parity of the step): K % 64 == 0, other K run the kernel above.  Arithmetic per
// output element unchanged.  ABL (measurement): 1 = no output stores, 2 = no loads after the prologue.
#define GD_WSTAGE 1536                    // units per W stage (one packed tile image)
#define GD_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))      /* vmcnt(n) only */
template <int FLAGS, int ABL>
__global__ void __launch_bounds__(256, 2) gemm_bf16x3_d_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, unsigned wp_bytes,
                                                               const float *__restrict__ bias, float *__restrict__ Y, int ldy, int M, int N, int K, int nnb, int nmb)
{
    __shared__ uint4 lds[3 * GB_TERM_UNITS + 2 * GD_WSTAGE];
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
    const int brow = 3 * GB_TERM_UNITS + (lane >> 5) * 128 + 64 * wn + (lane & 31);
    const unsigned rowb = (unsigned)ldy * 4u;
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, bias ? N * 4 : 0, GB_RSRC_FLAGS);

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
        const unsigned img = (unsigned)(nbw * nkb + wk) * (unsigned)(GD_WSTAGE * 16);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const unsigned so = __builtin_amdgcn_readfirstlane(img + (unsigned)(64 * (wid + 4 * q)) * 16u);
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + 16u * (unsigned)(3 * GB_TERM_UNITS + stage * GD_WSTAGE + 64 * (wid + 4 * q)));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(lane16), "s"(wdesc), "s"(so) : "memory");
        }
        if (++wk == nkb) { wk = 0; wj = next_item(wj); }
    };
    // X stream (registers, two steps ahead)
    const float *lx0, *lx1;
    int lk = 0, lj = j;
    auto x_tile_start = [&](int jj) {
        GD_TILE(jj, mb, nb); (void)nb;
        lx0 = X + (size_t)min(mb * GB_BM + xr[0], M - 1) * ldx + xk[0];
        lx1 = X + (size_t)min(mb * GB_BM + xr[1], M - 1) * ldx + xk[1];
    };
    float4 xa0, xa1, xb0, xb1, xc0, xc1, xd0, xd1;         // set 0: xa (rows u >> 2), xb (+ 64 rows); set 1: xc, xd
#define GD_XLOAD(p0, p1, q0, q1) do { \
        p0 = *(const float4 *)lx0; p1 = *(const float4 *)(lx0 + 4); q0 = *(const float4 *)lx1; q1 = *(const float4 *)(lx1 + 4); \
        lx0 += GB_BK; lx1 += GB_BK; \
        if (++lk == nkb) { lk = 0; lj = next_item(lj); x_tile_start(lj); } } while (0)
    auto xsplit_store = [&](const float4 &p, const float4 &q, int dst) {
        const float x[8] = { p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w };
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gb_split3(x[e], h[e], m[e], l[e]);
        lds[0 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(h[0], h[1]), gb_pack(h[2], h[3]), gb_pack(h[4], h[5]), gb_pack(h[6], h[7]));
        lds[1 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(m[0], m[1]), gb_pack(m[2], m[3]), gb_pack(m[4], m[5]), gb_pack(m[6], m[7]));
        lds[2 * GB_TERM_UNITS + dst] = make_uint4(gb_pack(l[0], l[1]), gb_pack(l[2], l[3]), gb_pack(l[4], l[5]), gb_pack(l[6], l[7]));
    };

    f32x16 acc[2][2];
    unsigned ov[2][2][16];                                 // FLAGS & 2: the tile of Y, fetched during the last K step
#define GD_SOFF(i, r) ((unsigned)(32 * (i) + ((r) & 3) + 8 * ((r) >> 2)) * rowb)
#define GD_PROD(ta, tb) do { \
        acc[0][0] = GB_MFMA(a[0][ta].v, b[0][tb].v, acc[0][0]); acc[0][1] = GB_MFMA(a[0][ta].v, b[1][tb].v, acc[0][1]); \
        acc[1][0] = GB_MFMA(a[1][ta].v, b[0][tb].v, acc[1][0]); acc[1][1] = GB_MFMA(a[1][ta].v, b[1][tb].v, acc[1][1]); } while (0)
    // step of parity P: X register set P -> the X stage; W stage P (filled by the DMA of the previous step) is multiplied; the DMA of the next
    // step's W goes to stage P ^ 1 and set P is reloaded with the X of two steps ahead.  Younger than the DMA this step waits for: the previous
    // step's 4 X loads, this step's 6 DMAs and 4 X loads.
#define GD_STEP(P, p0, p1, q0, q1, LAST) do { \
        __syncthreads(); \
        xsplit_store(p0, p1, xdst[0]); xsplit_store(q0, q1, xdst[1]); \
        if (ABL != 2) { wdma((P) ^ 1); GD_XLOAD(p0, p1, q0, q1); } \
        if ((LAST) && (FLAGS & 2)) { \
            _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) \
                        ov[i][jj][r] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff + 128u * jj, GD_SOFF(i, r), 0); \
            GD_VMCNT(63); } \
        else GD_VMCNT(14); \
        __syncthreads(); \
        __builtin_amdgcn_sched_barrier(0); \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
            GbFrag a[2][3], b[2][3]; \
            _Pragma("unroll") for (int t = 0; t < 3; ++t) \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
                    a[i][t].q = lds[t * GB_TERM_UNITS + 2 * ks * GB_KG_STRIDE + arow + 32 * i]; \
                    b[i][t].q = lds[(P) * GD_WSTAGE + t * 512 + 2 * ks * 128 + brow + 32 * i]; } \
            GD_PROD(1, 1); GD_PROD(0, 2); GD_PROD(2, 0); GD_PROD(0, 1); GD_PROD(1, 0); GD_PROD(0, 0); } } while (0)

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
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
        for (int kb = 0; kb < nkb - 2; kb += 2) { GD_STEP(0, xa0, xa1, xb0, xb1, false); GD_STEP(1, xc0, xc1, xd0, xd1, false); }
        GD_STEP(0, xa0, xa1, xb0, xb1, false);
        GD_STEP(1, xc0, xc1, xd0, xd1, true);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (n0 + 32 * jj >= N) continue;
            const float bv = jj ? bv1 : bv0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][jj][r] + bv;
                    if (FLAGS & 1) v = fmaxf(v, 0.f);
                    if (FLAGS & 2) v += __builtin_bit_cast(float, ov[i][jj][r]);
                    if (ABL == 1 && v != 123456.789f) continue;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, yoff + 128u * jj, GD_SOFF(i, r), 0);
                }
        }
        const int jn = next_item(j);
        if (jn == j) break;
        j = jn;
    }
    GD_VMCNT(0);                                           // the streams' last (unused) DMA must not land in the LDS of the next workgroup
#undef GD_STEP
#undef GD_PROD
#undef GD_SOFF
#undef GD_XLOAD
#undef GD_TILE
}

// ---- round 4 (flag 32; not the default -- it measured no faster): 256 x 128 tiles, eight wavefronts, two LDS stages, the two wavefronts of a SIMD in OPPOSITE phase --------------------
// tools/ablate_gemm.py on the kernels above: removing the output stores changes nothing (150 -> 147 us), so the loss is inside the K loop --
// a K step costs a SIMD ~5700 cycles for 2 x 48 MFMAs (3072 cycles).  With one LDS stage a step is [split + store | barrier | read +
// multiply | barrier]; the two workgroups of a CU drift into the same phase, contend for the LDS store path and then for the matrix core,
// and the step takes the SUM of the phases.  Here one workgroup of eight wavefronts (two per SIMD) owns the CU, LDS holds two stages
// (2 x 74 KB), and a step is [multiply stage k] + [split + store stage k + 1, issue the loads of k + 2] in EITHER order followed by one
// barrier: wavefronts 0-3 store first, wavefronts 4-7 (the SIMDs' second wavefronts) multiply first, so every SIMD always has one wavefront
// feeding the matrix core while the other one splits.  W tiles are the same packed images (two row blocks share each), arithmetic per
// output element unchanged (bitwise equal to the kernels above).  ABL (measurement): 1 = no output stores, 2 = all wavefronts store first.
#define G8_BM 256
#define G8_XKG 257                        // 16-byte units per k group of the X stage (256 rows + 1 pad)
#define G8_XTERM (4 * G8_XKG)
#define G8_WTERM (4 * GB_KG_STRIDE)
#define G8_STAGE (3 * G8_XTERM + 3 * G8_WTERM)   // 4632 units = 74112 bytes
template <int FLAGS, int ABL>
__global__ void __launch_bounds__(512, 1) gemm_bf16x3_w8_kernel(const float *__restrict__ X, int ldx, const uint4 *__restrict__ Wp, const float *__restrict__ bias,
                                                                float *__restrict__ Y, int ldy, int M, int N, int K, int nnb, int nmb)
{
    __shared__ uint4 lds[2 * G8_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const bool sfirst = ABL == 2 || !(wid & 4);
    const int nkb = K / GB_BK;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int items = ((nmb + 7) >> 3) * nnb;
#define G8_TILE(j, mb_, nb_) const int nb_ = (j) % nnb, mb_ = ((j) / nnb) * 8 + xcd
    int j = slot;
    if (j >= items) return;
    { G8_TILE(j, mb, nb); (void)nb; if (mb >= nmb) return; }

    // staging: X unit (row, k group) = (tid >> 2 [+ 128], tid & 3): 8 floats; W: unit tid of each of the three term images
    const int xr0 = tid >> 2, xk = 8 * (tid & 3);
    const int xdst = (tid & 3) * G8_XKG + xr0;
    const int wdst = 3 * G8_XTERM + (tid >> 7) * GB_KG_STRIDE + (tid & 127);
    const int arow = (lane >> 5) * G8_XKG + 64 * wm + (lane & 31);
    const int brow = 3 * G8_XTERM + (lane >> 5) * GB_KG_STRIDE + 64 * wn + (lane & 31);
    const unsigned rowb = (unsigned)ldy * 4u;
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void *)bias, 0, bias ? N * 4 : 0, GB_RSRC_FLAGS);

    const float *lx0, *lx1;
    const uint4 *lw;
    int lk, lj = j;
    auto load_tile_start = [&](int jj) {
        G8_TILE(jj, mb, nb);
        lx0 = X + (size_t)min(mb * G8_BM + xr0, M - 1) * ldx + xk;
        lx1 = X + (size_t)min(mb * G8_BM + 128 + xr0, M - 1) * ldx + xk;
        lw = Wp + (size_t)nb * nkb * GB_W_TILE_UNITS + tid;
        lk = 0;
    };
    float4 xa0, xa1, xb0, xb1;
    uint4 w0, w1, w2;
#define G8_LOAD() do { \
        xa0 = *(const float4 *)lx0; xa1 = *(const float4 *)(lx0 + 4); xb0 = *(const float4 *)lx1; xb1 = *(const float4 *)(lx1 + 4); \
        w0 = lw[0]; w1 = lw[512]; w2 = lw[1024]; \
        lx0 += GB_BK; lx1 += GB_BK; lw += GB_W_TILE_UNITS; \
        if (++lk == nkb) { int jn = lj + per_xcd; if (jn < items) { G8_TILE(jn, mbn_, nbn_); (void)nbn_; if (mbn_ >= nmb) jn = items; } \
                           if (jn < items) lj = jn; load_tile_start(lj); } } while (0)
    auto xsplit_store = [&](const float4 &p, const float4 &q, int dst) {
        const float x[8] = { p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w };
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gb_split3(x[e], h[e], m[e], l[e]);
        lds[0 * G8_XTERM + dst] = make_uint4(gb_pack(h[0], h[1]), gb_pack(h[2], h[3]), gb_pack(h[4], h[5]), gb_pack(h[6], h[7]));
        lds[1 * G8_XTERM + dst] = make_uint4(gb_pack(m[0], m[1]), gb_pack(m[2], m[3]), gb_pack(m[4], m[5]), gb_pack(m[6], m[7]));
        lds[2 * G8_XTERM + dst] = make_uint4(gb_pack(l[0], l[1]), gb_pack(l[2], l[3]), gb_pack(l[4], l[5]), gb_pack(l[6], l[7]));
    };
#define G8_S(base) do { xsplit_store(xa0, xa1, (base) + xdst); xsplit_store(xb0, xb1, (base) + xdst + 128); \
        lds[(base) + wdst] = w0; lds[(base) + wdst + G8_WTERM] = w1; lds[(base) + wdst + 2 * G8_WTERM] = w2; } while (0)

    f32x16 acc[2][2];
    unsigned ov[2][2][16];                                 // FLAGS & 2: the tile of Y, fetched during the last K step
#define G8_SOFF(i, r) ((unsigned)(32 * (i) + ((r) & 3) + 8 * ((r) >> 2)) * rowb)
#define G8_PROD(ta, tb) do { \
        acc[0][0] = GB_MFMA(a[0][ta].v, b[0][tb].v, acc[0][0]); acc[0][1] = GB_MFMA(a[0][ta].v, b[1][tb].v, acc[0][1]); \
        acc[1][0] = GB_MFMA(a[1][ta].v, b[0][tb].v, acc[1][0]); acc[1][1] = GB_MFMA(a[1][ta].v, b[1][tb].v, acc[1][1]); } while (0)
#define G8_M(base) do { \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
            GbFrag a[2][3], b[2][3]; \
            _Pragma("unroll") for (int t = 0; t < 3; ++t) \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) { \
                    a[i][t].q = lds[(base) + t * G8_XTERM + 2 * ks * G8_XKG + arow + 32 * i]; \
                    b[i][t].q = lds[(base) + t * G8_WTERM + 2 * ks * GB_KG_STRIDE + brow + 32 * i]; } \
            G8_PROD(1, 1); G8_PROD(0, 2); G8_PROD(2, 0); G8_PROD(0, 1); G8_PROD(1, 0); G8_PROD(0, 0); } } while (0)
#define G8_YFETCH() do { \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) \
                    ov[i][jj][r] = __builtin_amdgcn_raw_buffer_load_b32(ry, yoff + 128u * jj, G8_SOFF(i, r), 0); } while (0)
    // one K step: multiply stage `cur`, fill stage `cur ^ 1` with the next step's operands (already in registers), fetch the step after that
#define G8_STEP(LAST) do { \
        const int cur = (g & 1) * G8_STAGE, nxt = G8_STAGE - cur; \
        if (sfirst) { \
            G8_S(nxt); G8_LOAD(); \
            if ((LAST) && (FLAGS & 2)) G8_YFETCH(); \
            __builtin_amdgcn_sched_barrier(0); \
            G8_M(cur); \
        } else { \
            if ((LAST) && (FLAGS & 2)) G8_YFETCH(); \
            G8_M(cur); \
            __builtin_amdgcn_sched_barrier(0); \
            G8_S(nxt); G8_LOAD(); \
        } \
        __syncthreads(); ++g; } while (0)

    int g = 0;
    load_tile_start(j);
    G8_LOAD();
    G8_S(0);
    G8_LOAD();
    __syncthreads();
    for (;;) {
        G8_TILE(j, mb, nb);
        const int m0 = mb * G8_BM;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(Y + (size_t)m0 * ldy), 0, (int)((unsigned)min(G8_BM, M - m0) * rowb), GB_RSRC_FLAGS);
        const int n0 = nb * GB_BN + 64 * wn + (lane & 31);
        const unsigned yoff = (unsigned)(64 * wm + 4 * (lane >> 5)) * rowb + 4u * (unsigned)n0;
        const float bv0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0, 0, 0));
        const float bv1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, 4u * (unsigned)n0 + 128u, 0, 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
        for (int kb = 0; kb < nkb - 1; ++kb) G8_STEP(false);
        G8_STEP(true);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (n0 + 32 * jj >= N) continue;
            const float bv = jj ? bv1 : bv0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][jj][r] + bv;
                    if (FLAGS & 1) v = fmaxf(v, 0.f);
                    if (FLAGS & 2) v += __builtin_bit_cast(float, ov[i][jj][r]);
                    if (ABL == 1 && v != 123456.789f) continue;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, yoff + 128u * jj, G8_SOFF(i, r), 0);
                }
        }
        int jn = j + per_xcd;
        if (jn >= items) break;
        { G8_TILE(jn, mbn, nbn); (void)nbn; if (mbn >= nmb) break; }
        j = jn;
    }
#undef G8_STEP
#undef G8_YFETCH
#undef G8_M
#undef G8_PROD
#undef G8_SOFF
#undef G8_S
#undef G8_LOAD
#undef G8_TILE
}

extern "C" {

size_t mfr_gemm_bf16x3_pack_bytes(int N, int K)
{
    if (N <= 0 || K <= 0 || (K % GB_BK)) return 0;
    return (size_t)((N + GB_BN - 1) / GB_BN) * (K / GB_BK) * GB_W_TILE_UNITS * 16;
}

int mfr_gemm_bf16x3_pack(const float *w, int N, int K, void *packed, void *stream)
{
    if (!w || !packed || N <= 0 || K <= 0 || (K % GB_BK)) return MFR_E_ARG;
    const long long total = (long long)(mfr_gemm_bf16x3_pack_bytes(N, K) / 16);
    hipLaunchKernelGGL(gb_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K, total, (uint4 *)packed);
    CHECK_LAUNCH();
    return 0;
}

int mfr_gemm_bf16x3(const float *x, int ldx, const void *packed_w, const float *bias, float *y, int ldy, int M, int N, int K, int flags, void *stream)
{
    // flags: 1 = ReLU, 2 = accumulate; A/B and the bitwise-agreement test: 4 = one tile per workgroup (the round-3 kernel), 8 / 16 = persistent
    // 128 x 128 workgroups without / with deferred stores; 256, 512 = ablation 1, 2 (measurement only)
    // 32 = the eight-wavefront 256 x 128 kernel; no variant flag: W by LDS-DMA (K % 64 == 0, packed weight < 4 GB), else as flag 8
    int pk = flags & 24;
    const int f = flags & 3, one_tile = flags & 4, w8 = flags & 32, abl = (flags >> 8) & 3;
    if (!one_tile && !pk && !w8 && ((K % 64) || (size_t)((N + GB_BN - 1) / GB_BN) * (K / GB_BK) * GB_W_TILE_UNITS * 16 >= 0xffffffffull)) pk = 8;
    if (!x || !packed_w || !y || M <= 0 || N <= 0 || K <= 0 || (K % GB_BK) || (ldx & 3) || ldx < K || ldy < N || flags < 0 || (flags & ~0x33f)) return MFR_E_ARG;
    if (((uintptr_t)x & 15)) return MFR_E_ARG;
    const int nnb = (N + GB_BN - 1) / GB_BN, nmb = (M + GB_BM - 1) / GB_BM;
    const long long tiles = (long long)nmb * nnb;
    if (tiles > 0x7fffffffll) return MFR_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (one_tile) {
#define GB_GO(F) hipLaunchKernelGGL((gemm_bf16x3_kernel<F>), dim3((unsigned)tiles), dim3(256), 0, st, x, ldx, (const uint4 *)packed_w, bias, y, ldy, M, N, K, nnb)
        switch (f) { case 0: GB_GO(0); break; case 1: GB_GO(1); break; case 2: GB_GO(2); break; default: GB_GO(3); break; }
#undef GB_GO
    } else if (pk) {
        // 2 workgroups per CU on 256 CUs; fewer when there are fewer tiles (multiple of 8: one share per XCD)
        const long long per_xcd = (long long)((nmb + 7) / 8) * nnb;
        const unsigned grid = 8u * (unsigned)(per_xcd < 64 ? per_xcd : 64);
#define GB_GO(F, D, A) hipLaunchKernelGGL((gemm_bf16x3_pk_kernel<F, D, A>), dim3(grid), dim3(256), 0, st, x, ldx, (const uint4 *)packed_w, bias, y, ldy, M, N, K, nnb, nmb)
#define GB_SW(D, A) switch (f) { case 0: GB_GO(0, D, A); break; case 1: GB_GO(1, D, A); break; case 2: GB_GO(2, D, A); break; default: GB_GO(3, D, A); break; }
        if (abl == 1) { if (pk == 8) { GB_SW(0, 1) } else { GB_SW(1, 1) } }
        else if (abl == 2) { GB_SW(0, 2) }
        else if (abl == 3) { GB_SW(0, 3) }
        else     { if (pk == 8) { GB_SW(0, 0) } else { GB_SW(1, 0) } }
#undef GB_SW
#undef GB_GO
    } else if (w8) {
        // one workgroup of eight wavefronts per CU
        const int nmb8 = (M + G8_BM - 1) / G8_BM;
        const long long per_xcd = (long long)((nmb8 + 7) / 8) * nnb;
        const unsigned grid = 8u * (unsigned)(per_xcd < 32 ? per_xcd : 32);
#define GB_GO(F, A) hipLaunchKernelGGL((gemm_bf16x3_w8_kernel<F, A>), dim3(grid), dim3(512), 0, st, x, ldx, (const uint4 *)packed_w, bias, y, ldy, M, N, K, nnb, nmb8)
#define GB_SW(A) switch (f) { case 0: GB_GO(0, A); break; case 1: GB_GO(1, A); break; case 2: GB_GO(2, A); break; default: GB_GO(3, A); break; }
        if (abl == 1) { GB_SW(1) } else if (abl == 2) { GB_SW(2) } else { GB_SW(0) }
#undef GB_SW
#undef GB_GO
    } else {
        const long long per_xcd = (long long)((nmb + 7) / 8) * nnb;
        const unsigned grid = 8u * (unsigned)(per_xcd < 64 ? per_xcd : 64);
        const unsigned wp_bytes = (unsigned)((size_t)nnb * (K / GB_BK) * GB_W_TILE_UNITS * 16);
#define GB_GO(F, A) hipLaunchKernelGGL((gemm_bf16x3_d_kernel<F, A>), dim3(grid), dim3(256), 0, st, x, ldx, (const uint4 *)packed_w, wp_bytes, bias, y, ldy, M, N, K, nnb, nmb)
#define GB_SW(A) switch (f) { case 0: GB_GO(0, A); break; case 1: GB_GO(1, A); break; case 2: GB_GO(2, A); break; default: GB_GO(3, A); break; }
        if (abl == 1) { GB_SW(1) } else if (abl == 2) { GB_SW(2) } else { GB_SW(0) }
#undef GB_SW
#undef GB_GO
    }
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
