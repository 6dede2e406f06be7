// winograd_bf16x3.hip -- 3x3 / stride 1 / pad 1 convolution (NCHW f32 in, f32 out) as a fused Winograd F(2x2, 3x3) kernel on the
// gfx950 BF16 matrix cores at fp32 accuracy ("bf16x3"), with the layer epilogue (+bias, activation, optional 2x2 max-pool,
// optional residual) folded into the output transform.
//
// Reference call site: SuperGlue_matcher / LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-120) -> the un-vendored
// SuperPoint encoder (conv1b..conv4b, convPa, convDa) and LoFTR ResNet-FPN backbone.
//
// Arithmetic.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A per (cin, cout): 16 independent GEMMs over cin, one per Winograd position
// (i, j).  Every fp32 operand of those GEMMs -- the transformed filter U = G g G^T and the transformed patch V = B^T d B, both
// computed in fp32 exactly as the exact-fp32 kernel (winograd_conv.hip) does -- is split EXACTLY into three bf16 terms
// x = h + m + l (truncation, 8 + 8 + 8 significand bits) and a product u.v is evaluated as the six partial products
// hh + hm + mh + hl + lh + mm, each exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms are below
// 2^-26 |u||v|.  Against an fp64 product this has the error of the exact-fp32 MFMA (rms 2.4e-8 vs 2.8e-8 of sum|u||v|,
// tools/ubench/bf16x3_probe.hip -> profiles/r03_bf16x3_probe.jsonl) at 16/6 = 2.7x its matrix rate -- and the bf16 MFMA does
// not occupy the vector ALU the way the f32 MFMA does.
//
// Mapping to CDNA4 (what the numbers force):
//   * the split costs 5.5 VALU per V element, so V must be produced ONCE per (tile, cin) for all output channels of the
//     workgroup, and bf16 MFMAs eat operands 2.7x faster than the f32 ones, so neither operand may be re-read through LDS:
//     wavefront i (0..3) of a workgroup owns Winograd ROW i -- the four positions (i, 0..3) -- for 64 tiles x 64 output channels.
//     Row i of B^T d needs only two of the four patch rows, its four V values per (tile, cin) are produced and split in
//     registers and ARE the B operands of that wavefront's MFMAs; the filter fragments of row i are used by that wavefront
//     only and stream from L2 straight into registers (pre-split, pre-packed in operand order).  No operand crosses LDS.
//   * accumulators: 4 positions x (2 x 32 couts) x (2 x 32 tiles) = 16 MFMA tiles = 256 registers (the AGPR half), one
//     wavefront per SIMD, one workgroup per CU.  Per K step (16 input channels) and wavefront: 96 MFMAs, 24 filter
//     fragments (1 KB each), 480 VALU (64 row combinations, 64 column combinations, 352 split / pack).
//   * the raw input patches (10 rows x 34 columns x 16 channels per K step) are the only thing staged in LDS: 16-byte
//     buffer loads (8 per wavefront and K step, requested one step ahead, zero padding applied in registers) -> ds_write_b128,
//     double buffered, one barrier per K step; the staged row stride (48 floats) puts the two tile rows a wavefront reads
//     together on disjoint banks.  (LDS-DMA was tried first: its per-instruction issue cost beside the MFMA stream --
//     40 dword transfers per wavefront and K step -- was a third of the kernel's time.)
//   * the output transform is linear: each wavefront reduces its row over j in registers (2 values per tile and channel),
//     the four rows meet once through LDS, and wavefront q finishes one (32-cout, 32-tile) quarter: bias, activation,
//     max-pool / residual, coalesced stores.
//   * grid: 1-D, XCD-aware (every XCD walks one contiguous raster range of spatial blocks, cout groups innermost).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float wb_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float wb_lds_f32;
typedef __attribute__((address_space(3))) wb_f32x2 wb_lds_f32x2;

#define WB_RSRC_FLAGS 0x00020000
#define WB_OOB 0x80000000u
#define WB_ROWS 10                 // input rows per workgroup (4 tile rows: 8 output rows + 2 halo)
#define WB_RS 48                   // staged row stride (floats): 2 rows = 96 dwords = half the LDS banks apart
#define WB_CH (WB_ROWS * WB_RS)    // floats per staged channel
#define WB_STAGE (16 * WB_CH)      // floats per staged K step: [16 cin][10 rows][48]
#define WB_FRAGS_PER_KSTEP 96      // 4 i x 4 j x 2 cout blocks x 3 terms

union WbFrag { bf16x8 v; unsigned u[4]; uint4 q; };

__device__ __forceinline__ void wb_split3(float x, unsigned &h, unsigned &m, unsigned &l)
{
    // the upper 16 bits of each word are the bf16 term; h + m + l == x exactly
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned wb_pack(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// ---------------------------------------------------------------------------------------------------------------------
// filters: w [Cout, Cin, 3, 3] f32 -> U = G g G^T (fp64 arithmetic, rounded once to fp32), split into three bf16 terms and
// packed as MFMA A operands: fragment f = ((((cg * nks + c) * 4 + i) * 4 + j) * 2 + mb) * 3 + term, 64 lanes x 16 bytes;
// lane l holds cout cg*64 + mb*32 + (l & 31), input channels 16 c + 8 (l >> 5) + (0..7).  Channels beyond Cin / Cout are zero.
__global__ void __launch_bounds__(256) wb_filter_kernel(const float *__restrict__ w, int Cin, int Cout, int nks, long long total, uint4 *__restrict__ upk)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int l = (int)(t & 63);
    long long f = t >> 6;
    const int term = (int)(f % 3); f /= 3;
    const int mb = (int)(f & 1); f >>= 1;
    const int j = (int)(f & 3); f >>= 2;
    const int i = (int)(f & 3); f >>= 2;
    const int c = (int)(f % nks);
    const int cg = (int)(f / nks);
    const int co = cg * 64 + mb * 32 + (l & 31);
    const double G[4][3] = { { 1.0, 0.0, 0.0 }, { 0.5, 0.5, 0.5 }, { 0.5, -0.5, 0.5 }, { 0.0, 0.0, 1.0 } };
    unsigned word[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = 16 * c + 8 * (l >> 5) + e;
        float u = 0.f;
        if (co < Cout && ci < Cin) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) s += G[i][a] * (double)g[3 * a + b] * G[j][b];
            u = (float)s;
        }
        unsigned h, m, lo;
        wb_split3(u, h, m, lo);
        word[e] = term == 0 ? h : term == 1 ? m : lo;
    }
    upk[t] = make_uint4(wb_pack(word[0], word[1]), wb_pack(word[2], word[3]), wb_pack(word[4], word[5]), wb_pack(word[6], word[7]));
}

// ---------------------------------------------------------------------------------------------------------------------
#define WB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ABL & 16: s_memtime stamps of one workgroup's four wavefronts (tools/ablate_conv_bf16x3.py --profile)
__device__ unsigned long long wb_prof[4][64];
#define WB_STAMP(k) do { if ((ABL & 16) && blockIdx.x == prof_wg && lane == 0) { __builtin_amdgcn_sched_barrier(0); wb_prof[wi][(k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)

// ABL != 0: timing ablations for tools/ablate_conv_bf16x3.py (results are WRONG): 1 = no output transform / exchange / stores,
// 2 = no filter fragment loads, 4 = no V production (column combinations, split), 8 = no patch staging (loads, LDS)
template <bool POOL, int ABL = 0>
__global__ void __launch_bounds__(256, 1) wino_bf16x3_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ upk, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncg, int nks, int act)
{
    // one array: [2][WB_STAGE] patch staging during the K loop (60 KB), the 4 x 32 KB row partials afterwards (128 KB)
    __shared__ __attribute__((aligned(16))) float lds[32768];
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int cg = jj % ncg;
    const int sl = jj / ncg;
    const int s = xcd * Sx + sl;
    if (sl >= Sx || s >= S) return;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);           // Winograd row owned by this wavefront
    const int col = lane & 15, tysub = (lane >> 4) & 1, kg = lane >> 5;
    const int HW = H * W;
    const unsigned prof_wg = (gridDim.x / 2) | 5u;
    WB_STAMP(0);

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)b * Cin * HW), 0, Cin * HW * 4, WB_RSRC_FLAGS);
    // ---- patch staging: per K step and channel a 10-row x 48-column window (columns 32 bx - 1 + cx, rows 8 by - 1 + r; 34 x 10
    // are used), fetched as 16-byte pieces: piece = lane (60 of 64 lanes), 5 rows x 12 pieces per instruction, two instructions
    // per channel, wavefront wi stages channels 4 wi .. 4 wi + 3 of the 16.  Everything outside the image becomes zero HERE
    // (whole piece out of range -> the buffer returns 0; pieces that straddle the left / right image border are patched in
    // registers), so the readers never test anything.
    const int prow = lane / 12, pk = lane - 12 * prow;                  // lanes 60..63: prow = 5 -> no piece
    const int pix = 32 * bx - 1 + 4 * pk;                                // first column of this lane's piece
    unsigned voff[2];
    bool fixl = false;
    unsigned keep = 0xfu;                                                // which of the 4 dwords lie inside the row
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int iy = 8 * by - 1 + 5 * h + prow;
        const bool ok = prow < 5 && pk < 9 && iy >= 0 && iy < H && pix < W && pix + 3 >= 0;
        voff[h] = ok ? (unsigned)(iy * W + max(pix, 0)) * 4u : WB_OOB;   // the piece that starts at column -1 is fetched from column 0 ...
    }
    if (pix < 0) fixl = true;                                            // ... and shifted right by one in registers
    {
        unsigned k = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) if (pix + d >= 0 && pix + d < W) k |= 1u << d;
        keep = k;
    }
    const bool edge = (bx == 0) || (32 * bx + 35 >= W);                  // wave-uniform: some piece of this workgroup straddles a border
    uint4 praw[8];                                                       // the K step's 8 pieces of this lane, in flight
    auto pload = [&](int c) {
        if (ABL & 8) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned so = (unsigned)(16 * c + 4 * wi + q) * (unsigned)HW * 4u;   // channels >= Cin lie beyond the buffer: zeros
#pragma unroll
            for (int h = 0; h < 2; ++h) praw[2 * q + h] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[h], so, 0));
        }
    };
    auto pstore = [&](int buf) {
        if (ABL & 8) return;
        if (edge) {                                            // workgroups that touch the left / right image border (wave-uniform)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint4 v = praw[i];
                if (fixl) v = make_uint4(0u, v.x, v.y, v.z);
                v.x = (keep & 1u) ? v.x : 0u; v.y = (keep & 2u) ? v.y : 0u; v.z = (keep & 4u) ? v.z : 0u; v.w = (keep & 8u) ? v.w : 0u;
                praw[i] = v;
            }
        }
        if (prow < 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *(uint4 *)(lds + buf * WB_STAGE + (4 * wi + q) * WB_CH + (5 * h) * WB_RS + lane * 4) = praw[2 * q + h];
        }
    };

    // row i of B^T d:  i = 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
    const int ra = (wi == 0) ? 0 : (wi == 2) ? 2 : 1;
    const int rb = (wi == 0) ? 2 : (wi == 2) ? 1 : (wi == 1) ? 2 : 3;
    const float sg = (wi == 1) ? 1.0f : -1.0f;
    // this lane's patch-row offsets (floats inside one staged channel) for tile block nb: rows 2 (2 nb + tysub) + {ra, rb},
    // columns 2 col .. 2 col + 3 of the window
    int ofa[2], ofb[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int tr = 2 * nb + tysub;
        ofa[nb] = (2 * tr + ra) * WB_RS + 2 * col;
        ofb[nb] = (2 * tr + rb) * WB_RS + 2 * col;
    }
    float wv[2][8][4];
    // One wavefront per SIMD: an LDS round trip is ~130 cycles that nothing else hides, so the reads go out eight patches at a
    // time (64 registers in flight) and only MFMAs may be scheduled across the batch boundaries.
    auto wread = [&](int buf, int c) {
        const float *st = lds + buf * WB_STAGE + (8 * kg) * WB_CH;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float2 ra[8][4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float *ch = st + e * WB_CH;
                if (ABL & 8) { ra[e][0] = make_float2(1.f + e, 2.f + c); ra[e][1] = make_float2(3.f + nb, 4.f + lane); ra[e][2] = ra[e][1]; ra[e][3] = ra[e][0]; }
                else {
                    ra[e][0] = *(const float2 *)(ch + ofa[nb]); ra[e][1] = *(const float2 *)(ch + ofa[nb] + 2);
                    ra[e][2] = *(const float2 *)(ch + ofb[nb]); ra[e][3] = *(const float2 *)(ch + ofb[nb] + 2);
                }
            }
            __builtin_amdgcn_sched_barrier(0x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wv[nb][e][0] = __builtin_fmaf(sg, ra[e][2].x, ra[e][0].x); wv[nb][e][1] = __builtin_fmaf(sg, ra[e][2].y, ra[e][0].y);
                wv[nb][e][2] = __builtin_fmaf(sg, ra[e][3].x, ra[e][1].x); wv[nb][e][3] = __builtin_fmaf(sg, ra[e][3].y, ra[e][1].y);
            }
            __builtin_amdgcn_sched_barrier(0x8);
        }
    };

    f32x16 acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mb][nb][r] = 0.f;

    const uint4 *ubase = upk + ((size_t)cg * nks * WB_FRAGS_PER_KSTEP + (size_t)wi * 24) * 64 + lane;
    // filter fragments of position (wi, j) of K step c: [mb][term]
    auto aload = [&](WbFrag (&af)[2][3], int c, int j) {
        const uint4 *uc = ubase + ((size_t)c * WB_FRAGS_PER_KSTEP + j * 6) * 64;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (ABL & 2) af[mb][t].q = make_uint4(0x3f803f80u + lane, 0x3f803f80u + j, 0x3f803f80u + t, 0x3f803f80u + c);
                else af[mb][t].q = uc[(mb * 3 + t) * 64];
            }
    };
    // V(wi, j) of this lane's 2 x 8 patches, split and packed: [nb][term]
    auto vmake = [&](WbFrag (&vf)[2][3], int j) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            if (ABL & 4) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) vf[nb][t].u[k] = __float_as_uint(wv[nb][2 * k][j]) ^ __float_as_uint(wv[nb][2 * k + 1][t]);
                continue;
            }
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float *q = wv[nb][e];
                const float v = (j == 0) ? q[0] - q[2] : (j == 1) ? q[1] + q[2] : (j == 2) ? q[2] - q[1] : q[1] - q[3];
                wb_split3(v, h[e], m[e], l[e]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vf[nb][0].u[k] = wb_pack(h[2 * k], h[2 * k + 1]);
                vf[nb][1].u[k] = wb_pack(m[2 * k], m[2 * k + 1]);
                vf[nb][2].u[k] = wb_pack(l[2 * k], l[2 * k + 1]);
            }
        }
    };
    // six partial products per accumulator, small terms first; the four accumulators of a position alternate
#define WB_FENCE() do { if (!(ABL & 32)) __builtin_amdgcn_sched_barrier(0); } while (0)
#define WB_PROD(j, af, vf, ta, tb) do { \
        acc[j][0][0] = WB_MFMA(af[0][ta].v, vf[0][tb].v, acc[j][0][0]); acc[j][1][0] = WB_MFMA(af[1][ta].v, vf[0][tb].v, acc[j][1][0]); \
        acc[j][0][1] = WB_MFMA(af[0][ta].v, vf[1][tb].v, acc[j][0][1]); acc[j][1][1] = WB_MFMA(af[1][ta].v, vf[1][tb].v, acc[j][1][1]); } while (0)
#define WB_PHASE(j, af, vf) do { WB_PROD(j, af, vf, 1, 1); WB_PROD(j, af, vf, 0, 2); WB_PROD(j, af, vf, 2, 0); \
                                 WB_PROD(j, af, vf, 0, 1); WB_PROD(j, af, vf, 1, 0); WB_PROD(j, af, vf, 0, 0); } while (0)

    // ---- prologue: stage 0 -> LDS, row combinations of step 0, operands of phase (0, 0)
    WbFrag afA[2][3], afB[2][3], vfA[2][3], vfB[2][3];
    pload(0);
    aload(afA, 0, 0);
    pstore(0);
    if (nks > 1) pload(1);
    __syncthreads();
    WB_STAMP(1);
    wread(0, 0);
    vmake(vfA, 0);
    WB_STAMP(2);

    // ---- K loop: one step = 4 phases (positions j = 0..3), each: operands of the NEXT phase requested / produced, then the 24
    // MFMAs of this one.  The patches of step c + 1 are requested in phase 0, written to LDS after phase 2, and turned into row
    // combinations under the MFMAs of phase 3 (the combinations of step c are dead by then).
    // The steady-state body has one wave-uniform branch (the border fix-up of the staged pieces, right before the barrier); the
    // last step is peeled.
    {
        int c = 0;
        // vmcnt retires in order: a (slow, HBM) patch request delays every (fast, L2) filter request issued after it.  So the
        // patches of step c + 2 are requested LAST in step c -- right before the barrier, ~1000 cycles ahead of the next filter
        // request and a whole step ahead of their own use -- and the loop never waits on them.
        for (; c + 1 < nks; ++c) {
            WB_STAMP(3 + 6 * c);
            // (WB_FENCE: the six filter requests of the next phase go out BEFORE anything else of this phase -- left alone, hipcc spreads
            // them between the phase's MFMAs and the late ones are waited for at the top of the next phase)
            aload(afB, c, 1); WB_FENCE(); vmake(vfB, 1);
            WB_PHASE(0, afA, vfA);
            WB_STAMP(4 + 6 * c);
            aload(afA, c, 2); WB_FENCE(); vmake(vfA, 2);
            WB_PHASE(1, afB, vfB);
            WB_STAMP(5 + 6 * c);
            aload(afB, c, 3); WB_FENCE(); vmake(vfB, 3);
            WB_PHASE(2, afA, vfA);
            WB_STAMP(6 + 6 * c);
            pstore((c + 1) & 1);                            // (moving this under phases 1-2 costs 5 spilled registers and measured 5 % slower)
            aload(afA, c + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            pload(c + 2);                                   // beyond the last step: channels >= Cin -> out of range, returns zeros
            WB_STAMP(7 + 6 * c);
            __syncthreads();
            WB_STAMP(8 + 6 * c);
            // (queueing half of phase 3 before the barrier, so that the matrix pipe works while the wavefronts meet, measured 6 % SLOWER:
            // the row combinations and the V split after the barrier then have 12 instead of 24 MFMAs to hide under)
            wread((c + 1) & 1, c + 1); vmake(vfA, 0);
            WB_PHASE(3, afB, vfB);
        }
        WB_STAMP(3 + 6 * c);
        aload(afB, c, 1); WB_FENCE(); vmake(vfB, 1);
        WB_PHASE(0, afA, vfA);
        aload(afA, c, 2); WB_FENCE(); vmake(vfA, 2);
        WB_PHASE(1, afB, vfB);
        aload(afB, c, 3); WB_FENCE(); vmake(vfB, 3);
        WB_PHASE(2, afA, vfA);
        WB_PHASE(3, afB, vfB);
    }
    WB_STAMP(40);
#undef WB_PHASE
#undef WB_PROD
#undef WB_FENCE

    if (ABL & 1) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[j][mb][nb][r];
        if (t == 12345.678f) y[tid] = t;
        return;
    }
    // ---- output transform.  Row partials over j (A^T applied along j): pa = M0 + M1 + M2, pb = M1 - M2 - M3; the four rows meet
    // in LDS: part[row][ab][mb][nb][r4][lane] (float4 = registers 4 r4 .. 4 r4 + 3); quarter (mb, nb) is finished by wavefront 2 mb + nb.
    // Straight-line code: one wavefront per SIMD hides nothing, so every load of a stage is requested before the first use.
    __builtin_amdgcn_sched_barrier(0);
    const int qmb = wi >> 1, qnb = wi & 1;
    const int co0 = cg * 64 + qmb * 32 + 4 * kg;            // this lane's channels: co0 + (r & 3) + 8 (r >> 2)
    __syncthreads();                                        // every wavefront is done with the patch stages
    WB_STAMP(41);
    float4 *part = (float4 *)lds;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * r4 + k;
                    pa[k] = (acc[0][mb][nb][r] + acc[1][mb][nb][r]) + acc[2][mb][nb][r];
                    pb[k] = (acc[1][mb][nb][r] - acc[2][mb][nb][r]) - acc[3][mb][nb][r];
                }
                part[((((wi * 2 + 0) * 2 + mb) * 2 + nb) * 4 + r4) * 64 + lane] = make_float4(pa[0], pa[1], pa[2], pa[3]);
                part[((((wi * 2 + 1) * 2 + mb) * 2 + nb) * 4 + r4) * 64 + lane] = make_float4(pb[0], pb[1], pb[2], pb[3]);
            }
    WB_STAMP(42);
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = 0.f;
    if (bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias[min(co0 + (r & 3) + 8 * (r >> 2), Cout - 1)];
    }
    __syncthreads();
    WB_STAMP(43);

    const int tr = 2 * qnb + tysub;
    const int ty = 4 * by + tr, tx = 16 * bx + col;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const float4 *pq = part + ((qmb * 2 + qnb) * 4) * 64 + lane;       // + (row * 2 + ab) * 16 * 64 + r4 * 64
    float Y[16][4];                                                    // y00, y01, y10, y11 per channel register
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        float4 P[2][4][2];                                             // [r4 - 2 hh][row][ab]: 16 reads in flight together
#pragma unroll
        for (int r4 = 0; r4 < 2; ++r4)
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) P[r4][row][ab] = pq[((row * 2 + ab) * 16 + 2 * hh + r4) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r4 = 0; r4 < 2; ++r4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * (2 * hh + r4) + k;
#define WB_EL(v) (k == 0 ? (v).x : k == 1 ? (v).y : k == 2 ? (v).z : (v).w)
                // A^T applied along i:  Y[0][.] = P0 + P1 + P2,  Y[1][.] = P1 - P2 - P3
                Y[r][0] = ((WB_EL(P[r4][0][0]) + WB_EL(P[r4][1][0])) + WB_EL(P[r4][2][0])) + bv[r];
                Y[r][1] = ((WB_EL(P[r4][0][1]) + WB_EL(P[r4][1][1])) + WB_EL(P[r4][2][1])) + bv[r];
                Y[r][2] = ((WB_EL(P[r4][1][0]) - WB_EL(P[r4][2][0])) - WB_EL(P[r4][3][0])) + bv[r];
                Y[r][3] = ((WB_EL(P[r4][1][1]) - WB_EL(P[r4][2][1])) - WB_EL(P[r4][3][1])) + bv[r];
#undef WB_EL
            }
        }
    }
    WB_STAMP(44);
    const size_t cstride = (size_t)Ho * Wo;
    float *yb = y + ((size_t)b * Cout + co0) * cstride;
    const bool allco = cg * 64 + 64 <= Cout;                          // wave-uniform: no padded output channels in this group
    if (POOL) {
        // wave-uniform decisions (activation code, "no padded channels in this group") are taken ONCE around straight-line loops:
        // a branch per store serialises sixteen store issues behind sixteen waits
        float m[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = fmaxf(fmaxf(Y[r][0], Y[r][1]), fmaxf(Y[r][2], Y[r][3]));   // the activations are monotone: act(max) = max(act)
        if (act == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = fmaxf(m[r], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = m[r] > 0.f ? m[r] : 0.01f * m[r];
        }
        if (ty < Ho && tx < Wo) {
            float *yo = yb + (size_t)ty * Wo + tx;
            if (allco) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yo[(size_t)((r & 3) + 8 * (r >> 2)) * cstride] = m[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dc = (r & 3) + 8 * (r >> 2);
                    if (co0 + dc < Cout) yo[(size_t)dc * cstride] = m[r];
                }
            }
        }
    } else {
        const int oy = 2 * ty, ox = 2 * tx;
        const bool c0 = ox < W, c1 = ox + 1 < W, r0 = oy < H, r1 = oy + 1 < H;
        if (residual) {
            const float *rb0 = residual + ((size_t)b * Cout + co0) * cstride + (size_t)oy * W + ox;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dc = (r & 3) + 8 * (r >> 2);
                if (allco || co0 + dc < Cout) {
                    const float *ro = rb0 + (size_t)dc * cstride;
                    if (r0 && c0) Y[r][0] += ro[0];
                    if (r0 && c1) Y[r][1] += ro[1];
                    if (r1 && c0) Y[r][2] += ro[W];
                    if (r1 && c1) Y[r][3] += ro[W + 1];
                }
            }
        }
        float *yo0 = yb + (size_t)oy * W + ox;
        if (act == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) Y[r][k] = fmaxf(Y[r][k], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) Y[r][k] = Y[r][k] > 0.f ? Y[r][k] : 0.01f * Y[r][k];
        }
        const bool interior = allco && r0 && r1 && c1 && !(W & 1);       // per lane; the common case: two float2 stores per channel
        if (interior) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *yo = yo0 + (size_t)((r & 3) + 8 * (r >> 2)) * cstride;
                *(float2 *)yo = make_float2(Y[r][0], Y[r][1]);
                *(float2 *)(yo + W) = make_float2(Y[r][2], Y[r][3]);
            }
        } else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float y00 = Y[r][0], y01 = Y[r][1], y10 = Y[r][2], y11 = Y[r][3];
            const int dc = (r & 3) + 8 * (r >> 2);
            if (!(allco || co0 + dc < Cout)) continue;
            float *yo = yo0 + (size_t)dc * cstride;
            if (c1) {
                if (!(W & 1)) {
                    if (r0) *(float2 *)yo = make_float2(y00, y01);
                    if (r1) *(float2 *)(yo + W) = make_float2(y10, y11);
                } else {
                    if (r0) { yo[0] = y00; yo[1] = y01; }
                    if (r1) { yo[W] = y10; yo[W + 1] = y11; }
                }
            } else if (c0) {
                if (r0) yo[0] = y00;
                if (r1) yo[W] = y10;
            }
        }
    }
    WB_STAMP(45);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the same arithmetic at TWO wavefronts per SIMD.  The kernel above owns the whole accumulator file (256 registers, one
// wavefront per SIMD): every LDS / L2 latency, the prologue and the output transform of a tile run with the matrix pipe idle
// (profiles/r03_ablate_conv_timeline.json: 45 k cycles per tile, 12.3 k of them MFMA).  Here a workgroup is still four wavefronts
// -- wavefront i = Winograd row i, so no operand is shared between wavefronts and none crosses LDS -- but it covers 64 tiles x
// 32 output channels: 128 accumulators, <= 256 registers, 64 KB of LDS, so TWO workgroups (two output-channel halves, or two
// spatial blocks) live on a CU and one's stalls, prologue and epilogue are covered by the other's MFMAs; no barrier couples them.
// What it costs: both workgroups of a tile produce the same V (the patch read, the row / column combinations and the 3-way
// split are done twice per CU: ~10 VALU per MFMA instead of 5), which the second wavefront's issue slots absorb.
// To fit 256 registers nothing is double-buffered in registers: per K step the 12 filter fragments of the row stay resident
// (48 registers, re-requested one by one right after their last use), the row combinations exist for ONE 32-tile half at a time
// (32 registers), V for one position at a time (12), and the patch pieces go to LDS by DMA (no register at all).
// ABL != 0: timing ablations (results are WRONG): 1 = no output transform / stores, 2 = no filter fragment loads, 4 = no V production
// (column combinations, split), 8 = no patch DMA and no LDS patch reads, 16 = no patch DMA only, 32 = no LDS patch reads only, 64 = no MFMAs.
// TUNE: experiments that keep the result right: 1 = no scheduling fences inside the V production, 2 = row combinations four channels at a time
template <bool POOL, int ABL = 0, int TUNE = 0>
__global__ void __launch_bounds__(256, 2) wino_bf16x3_w2_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ upk, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncg, int nks, int act)
{
    // [2][WB_STAGE] patch staging during the K loop (60 KB), the 4 x 16 KB row partials afterwards (64 KB)
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int cg = jj % ncg;                                              // 32-channel output group
    const int sl = jj / ncg;
    const int s = xcd * Sx + sl;
    if (sl >= Sx || s >= S) return;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, tysub = (lane >> 4) & 1, kg = lane >> 5;
    const int HW = H * W;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)b * Cin * HW), 0, Cin * HW * 4, WB_RSRC_FLAGS);
    // patch staging exactly as in the kernel above (same window, same piece <-> lane map, same border handling), one input channel
    // (two pieces per lane) at a time
    const int prow = lane / 12, pk = lane - 12 * prow;
    const int pix = 32 * bx - 1 + 4 * pk;
    unsigned voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int iy = 8 * by - 1 + 5 * h + prow;
        const bool ok = prow < 5 && pk < 9 && iy >= 0 && iy < H && pix < W && pix + 3 >= 0;
        voff[h] = ok ? (unsigned)(iy * W + max(pix, 0)) * 4u : WB_OOB;
    }
    const bool fixl = pix < 0;
    unsigned keep = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) if (pix + d >= 0 && pix + d < W) keep |= 1u << d;
    const bool edge = (bx == 0) || (32 * bx + 35 >= W);
    // LDS-DMA (buffer_load_dwordx4 ... lds): the staged layout is lane-linear per (channel, half) -- 5 rows x 12 pieces = 60 lanes x
    // 16 bytes -- so a piece goes from L2 / HBM to its LDS slot without touching a register, a whole K step ahead of its use.
    // Pieces outside the image come back as zeros (offset beyond the buffer); the few pieces that STRADDLE the left / right image
    // border are patched in LDS by the lane that requested them (pfix; edge workgroups only).
    // The request is inline asm on purpose: hipcc treats an LDS-DMA it knows about as a pending store to the whole LDS array and
    // waits vmcnt(0) before the next ds_read -- i.e. right after issuing it.  Nothing in the step reads the stage being filled;
    // the step's own vmcnt(0) (WB2_VMCNT0, a builtin hipcc does see) and the barrier order it before the next step's reads.
    typedef unsigned wb_u32x4 __attribute__((ext_vector_type(4)));
    wb_u32x4 xdesc;
    {
        const unsigned long long xa = (unsigned long long)(x + (size_t)b * Cin * HW);
        xdesc.x = __builtin_amdgcn_readfirstlane((unsigned)xa);
        xdesc.y = __builtin_amdgcn_readfirstlane((unsigned)(xa >> 32) & 0xffffu);
        xdesc.z = (unsigned)(Cin * HW) * 4u;
        xdesc.w = WB_RSRC_FLAGS;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    auto pdma = [&](int c, int buf) {
        if (ABL & (8 | 16)) return;
        if (prow < 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned so = (unsigned)(16 * c + 4 * wi + q) * (unsigned)HW * 4u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned m0v = lds0 + 4u * (unsigned)(buf * WB_STAGE + (4 * wi + q) * WB_CH + (5 * h) * WB_RS);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(voff[h]), "s"(xdesc), "s"(so) : "memory");
                }
            }
        }
    };
    auto pfix = [&](int buf) {
        if (edge && prow < 5 && (fixl || keep != 0xfu)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint4 *pp = (uint4 *)(lds + buf * WB_STAGE + (4 * wi + (k >> 1)) * WB_CH + (5 * (k & 1)) * WB_RS + lane * 4);
                uint4 v = *pp;
                if (fixl) v = make_uint4(0u, v.x, v.y, v.z);
                v.x = (keep & 1u) ? v.x : 0u; v.y = (keep & 2u) ? v.y : 0u; v.z = (keep & 4u) ? v.z : 0u; v.w = (keep & 8u) ? v.w : 0u;
                *pp = v;
            }
        }
    };
#define WB2_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)                  /* vmcnt(0) only; the builtin (not inline asm) so that hipcc's own wait bookkeeping sees it */

    const int ra = (wi == 0) ? 0 : (wi == 2) ? 2 : 1;
    const int rb = (wi == 0) ? 2 : (wi == 2) ? 1 : (wi == 1) ? 2 : 3;
    const float sg = (wi == 1) ? 1.0f : -1.0f;
    const int ofa0 = (2 * tysub + ra) * WB_RS + 2 * col, ofb0 = (2 * tysub + rb) * WB_RS + 2 * col;    // tile block nb: + 4 nb WB_RS
    float wv[8][4];
    // row combinations of one 32-tile half, two input channels at a time (register budget: 16 in flight, not 64)
    auto wread = [&](int buf, int nb) {
        const wb_lds_f32 *st = (const wb_lds_f32 *)lds + buf * WB_STAGE + (8 * kg) * WB_CH + 4 * nb * WB_RS;
        constexpr int EB = (TUNE & 2) ? 4 : 2;
#pragma unroll
        for (int e0 = 0; e0 < 8; e0 += EB) {
            wb_f32x2 r2[EB][4];
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const wb_lds_f32 *ch = st + (e0 + e) * WB_CH;
                if (ABL & (8 | 32)) { r2[e][0] = wb_f32x2{1.f + e0, 2.f + nb}; r2[e][1] = wb_f32x2{3.f + e, 4.f + lane}; r2[e][2] = r2[e][1]; r2[e][3] = r2[e][0]; continue; }
                // volatile: keeps hipcc from fusing the two 8-byte reads of a row into ds_read2_b64, which the LDS serves at a
                // quarter of the rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table: 16 vs 2 + 2 cycles per wavefront)
                r2[e][0] = *(const volatile wb_lds_f32x2 *)(ch + ofa0); r2[e][1] = *(const volatile wb_lds_f32x2 *)(ch + ofa0 + 2);
                r2[e][2] = *(const volatile wb_lds_f32x2 *)(ch + ofb0); r2[e][3] = *(const volatile wb_lds_f32x2 *)(ch + ofb0 + 2);
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                wv[e0 + e][0] = __builtin_fmaf(sg, r2[e][2].x, r2[e][0].x); wv[e0 + e][1] = __builtin_fmaf(sg, r2[e][2].y, r2[e][0].y);
                wv[e0 + e][2] = __builtin_fmaf(sg, r2[e][3].x, r2[e][1].x); wv[e0 + e][3] = __builtin_fmaf(sg, r2[e][3].y, r2[e][1].y);
            }
            // pin the combinations HERE: left alone, instruction selection sinks them below the next batches' reads and all 64 raw
            // values are live at once (the 20 registers that spilled)
#pragma unroll
            for (int e = 0; e < EB; e += 2)
                asm volatile("" : "+v"(wv[e0 + e][0]), "+v"(wv[e0 + e][1]), "+v"(wv[e0 + e][2]), "+v"(wv[e0 + e][3]),
                                  "+v"(wv[e0 + e + 1][0]), "+v"(wv[e0 + e + 1][1]), "+v"(wv[e0 + e + 1][2]), "+v"(wv[e0 + e + 1][3]) :: "memory");
        }
    };
    // V(wi, j) of the half, split and packed two input channels at a time
    auto vmake = [&](WbFrag (&vf)[3], int j) {
        if (ABL & 4) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) vf[t].u[k] = __float_as_uint(wv[2 * k][j]) ^ __float_as_uint(wv[2 * k + 1][t]);
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h[2], m[2], l[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float *q = wv[2 * k + e];
                const float v = (j == 0) ? q[0] - q[2] : (j == 1) ? q[1] + q[2] : (j == 2) ? q[2] - q[1] : q[1] - q[3];
                wb_split3(v, h[e], m[e], l[e]);
            }
            vf[0].u[k] = wb_pack(h[0], h[1]);
            vf[1].u[k] = wb_pack(m[0], m[1]);
            vf[2].u[k] = wb_pack(l[0], l[1]);
            if (!(TUNE & 1)) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // filter fragments of row wi: the packed layout keeps 64-channel groups [mb][term]; this workgroup's half is mb = cg & 1.
    // Buffer loads: the fragment index is wave-uniform (scalar offset), the only vector address is lane * 16.
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)upk, 0, (int)(((ncg + 1) >> 1) * nks * WB_FRAGS_PER_KSTEP * 1024), WB_RSRC_FLAGS);
    const unsigned fbase = (unsigned)(((cg >> 1) * nks * WB_FRAGS_PER_KSTEP + wi * 24 + (cg & 1) * 3) * 1024);
    const unsigned lane16 = (unsigned)lane * 16u;
    WbFrag F[4][3];
    auto aload = [&](int c, int j) {
        const unsigned so = fbase + (unsigned)((c * WB_FRAGS_PER_KSTEP + j * 6) * 1024);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (ABL & 2) F[j][t].q = make_uint4(0x3f803f80u + lane, 0x3f803f80u + j, 0x3f803f80u + t, 0x3f803f80u + c);
            else F[j][t].q = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsF, lane16, so + t * 1024u, 0));
        }
    };
#define WB2_PROD(j, nb, ta, tb) do { if (ABL & 64) { acc[j][nb][ta] += __uint_as_float(F[j][ta].u[tb] ^ vf[tb].u[ta]); } else acc[j][nb] = WB_MFMA(F[j][ta].v, vf[tb].v, acc[j][nb]); } while (0)
#define WB2_PHASE(j, nb) do { WB2_PROD(j, nb, 1, 1); WB2_PROD(j, nb, 0, 2); WB2_PROD(j, nb, 2, 0); \
                              WB2_PROD(j, nb, 0, 1); WB2_PROD(j, nb, 1, 0); WB2_PROD(j, nb, 0, 0); } while (0)

    // ---- prologue: the patches and the row's filter fragments of step 0
#pragma unroll
    for (int j = 0; j < 4; ++j) aload(0, j);
    pdma(0, 0);
    WB2_VMCNT0();
    pfix(0);
    __syncthreads();
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nb][r] = 0.f;

    // ---- K loop.  A step = 2 tile halves x 4 positions; sub-phase k = 4 nb + j: V(wi, j) of the half, then its six products.
    // Top of step c: the eight patch pieces of step c + 1 are requested (LDS-DMA into the other stage: a whole step to land);
    // the filter fragments of step c + 1 are requested right after the last product that reads those of step c; ONE vmcnt(0)
    // before the step's barrier covers both.  vmcnt retires in order and hipcc waits vmcnt(0) for a register load whenever an
    // LDS-DMA is in flight -- so no register load may be WAITED for between the DMA requests and that barrier: the fragments of
    // step c are complete (same vmcnt(0), one step earlier) before the DMAs of step c + 1 go out.
    int c = 0;
    for (; c + 1 < nks; ++c) {
        pdma(c + 1, (c + 1) & 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            wread(c & 1, nb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                WbFrag vf[3];
                vmake(vf, j);
                WB2_PHASE(j, nb);
                if (nb == 1) aload(c + 1, j);
            }
        }
        WB2_VMCNT0();
        pfix((c + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {                                      // last step: nothing left to request
        wread(c & 1, nb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            WbFrag vf[3];
            vmake(vf, j);
            WB2_PHASE(j, nb);
        }
    }
#undef WB2_PHASE
#undef WB2_PROD

    if (ABL & 1) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[j][nb][r];
        if (t == 12345.678f) y[tid] = t;
        return;
    }
    // ---- output transform: as above with one 32-channel block; the four rows meet in LDS as part[row][ab][nb][r4][lane] (float4 =
    // registers 4 r4 .. 4 r4 + 3); wavefront q finishes tile half q & 1, register groups r4 = 2 (q >> 1), 2 (q >> 1) + 1.
    const int qnb = wi & 1, qh = wi >> 1;
    const int co0 = cg * 32 + 4 * kg + 16 * qh;            // this lane's channels: co0 + (r' & 3) + 8 (r' >> 2), r' = 0..7
    __syncthreads();                                        // every wavefront is done with the patch stages
    float4 *part = (float4 *)lds;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            float pa[4], pb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * r4 + k;
                pa[k] = (acc[0][nb][r] + acc[1][nb][r]) + acc[2][nb][r];
                pb[k] = (acc[1][nb][r] - acc[2][nb][r]) - acc[3][nb][r];
            }
            part[(((wi * 2 + 0) * 2 + nb) * 4 + r4) * 64 + lane] = make_float4(pa[0], pa[1], pa[2], pa[3]);
            part[(((wi * 2 + 1) * 2 + nb) * 4 + r4) * 64 + lane] = make_float4(pb[0], pb[1], pb[2], pb[3]);
        }
    float bv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[r] = 0.f;
    if (bias) {
#pragma unroll
        for (int r = 0; r < 8; ++r) bv[r] = bias[min(co0 + (r & 3) + 8 * (r >> 2), Cout - 1)];
    }
    __syncthreads();

    const int tr = 2 * qnb + tysub;
    const int ty = 4 * by + tr, tx = 16 * bx + col;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const float4 *pq = part + (qnb * 4 + 2 * qh) * 64 + lane;          // + (row * 2 + ab) * 8 * 64 + r4' * 64
    float Y[8][4];
    {
        float4 P[2][4][2];
#pragma unroll
        for (int r4 = 0; r4 < 2; ++r4)
#pragma unroll
            for (int row = 0; row < 4; ++row)
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) P[r4][row][ab] = pq[((row * 2 + ab) * 8 + r4) * 64];
#pragma unroll
        for (int r4 = 0; r4 < 2; ++r4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * r4 + k;
#define WB_EL(v) (k == 0 ? (v).x : k == 1 ? (v).y : k == 2 ? (v).z : (v).w)
                Y[r][0] = ((WB_EL(P[r4][0][0]) + WB_EL(P[r4][1][0])) + WB_EL(P[r4][2][0])) + bv[r];
                Y[r][1] = ((WB_EL(P[r4][0][1]) + WB_EL(P[r4][1][1])) + WB_EL(P[r4][2][1])) + bv[r];
                Y[r][2] = ((WB_EL(P[r4][1][0]) - WB_EL(P[r4][2][0])) - WB_EL(P[r4][3][0])) + bv[r];
                Y[r][3] = ((WB_EL(P[r4][1][1]) - WB_EL(P[r4][2][1])) - WB_EL(P[r4][3][1])) + bv[r];
#undef WB_EL
            }
        }
    }
    const size_t cstride = (size_t)Ho * Wo;
    float *yb = y + ((size_t)b * Cout + co0) * cstride;
    const bool allco = cg * 32 + 32 <= Cout;
    if (POOL) {
        float m[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) m[r] = fmaxf(fmaxf(Y[r][0], Y[r][1]), fmaxf(Y[r][2], Y[r][3]));
        if (act == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) m[r] = fmaxf(m[r], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) m[r] = m[r] > 0.f ? m[r] : 0.01f * m[r];
        }
        if (ty < Ho && tx < Wo) {
            float *yo = yb + (size_t)ty * Wo + tx;
            if (allco) {
#pragma unroll
                for (int r = 0; r < 8; ++r) yo[(size_t)((r & 3) + 8 * (r >> 2)) * cstride] = m[r];
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int dc = (r & 3) + 8 * (r >> 2);
                    if (co0 + dc < Cout) yo[(size_t)dc * cstride] = m[r];
                }
            }
        }
    } else {
        const int oy = 2 * ty, ox = 2 * tx;
        const bool c0 = ox < W, c1 = ox + 1 < W, r0 = oy < H, r1 = oy + 1 < H;
        if (residual) {
            const float *rb0 = residual + ((size_t)b * Cout + co0) * cstride + (size_t)oy * W + ox;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int dc = (r & 3) + 8 * (r >> 2);
                if (allco || co0 + dc < Cout) {
                    const float *ro = rb0 + (size_t)dc * cstride;
                    if (r0 && c0) Y[r][0] += ro[0];
                    if (r0 && c1) Y[r][1] += ro[1];
                    if (r1 && c0) Y[r][2] += ro[W];
                    if (r1 && c1) Y[r][3] += ro[W + 1];
                }
            }
        }
        float *yo0 = yb + (size_t)oy * W + ox;
        if (act == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) Y[r][k] = fmaxf(Y[r][k], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) Y[r][k] = Y[r][k] > 0.f ? Y[r][k] : 0.01f * Y[r][k];
        }
        const bool interior = allco && r0 && r1 && c1 && !(W & 1);
        if (interior) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float *yo = yo0 + (size_t)((r & 3) + 8 * (r >> 2)) * cstride;
                *(float2 *)yo = make_float2(Y[r][0], Y[r][1]);
                *(float2 *)(yo + W) = make_float2(Y[r][2], Y[r][3]);
            }
        } else
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float y00 = Y[r][0], y01 = Y[r][1], y10 = Y[r][2], y11 = Y[r][3];
            const int dc = (r & 3) + 8 * (r >> 2);
            if (!(allco || co0 + dc < Cout)) continue;
            float *yo = yo0 + (size_t)dc * cstride;
            if (c1) {
                if (!(W & 1)) {
                    if (r0) *(float2 *)yo = make_float2(y00, y01);
                    if (r1) *(float2 *)(yo + W) = make_float2(y10, y11);
                } else {
                    if (r0) { yo[0] = y00; yo[1] = y01; }
                    if (r1) { yo[W] = y10; yo[W + 1] = y11; }
                }
            } else if (c0) {
                if (r0) yo[0] = y00;
                if (r1) yo[W] = y10;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4, second step: EIGHT wavefronts per workgroup, two per SIMD, nothing produced twice.
// What the counters say about the two kernels above (profiles/r04_pmc_conv1b_w2_vs_1wave.json): the SIMD issues ONE instruction
// per 4 cycles whatever the wavefront, a VALU instruction and an MFMA issue do not overlap, and the launch time is the SUM
// (VALU 63 % + matrix pipe 36 % of the SIMD's cycles in the two-workgroup kernel, 2703 VALU per 192 MFMAs per wavefront).  So
// the lever is the instruction count per tile, and what sets it is how often V -- 5.5 VALU per element for the exact 3-way split
// -- is produced.  Here a workgroup is the 64-tile x 64-channel block of the first kernel again, and wavefront w owns Winograd row
// i = w >> 1 and the column pair j in {2 jp, 2 jp + 1}, jp = w & 1: 2 positions x 2 channel blocks x 2 tile blocks = 128
// accumulators; every position's V is produced by exactly one wavefront (272 VALU per 48 MFMAs instead of 566), every filter
// fragment is requested by exactly one wavefront (12 per K step, resident, re-requested after their last use), the patches are
// staged once per workgroup by LDS-DMA (four 16-byte pieces per wavefront and K step).  Only the row combination B^T d of a row is
// evaluated by both wavefronts of a pair (48 of the 272).  The column pair decides which three of the four patch columns a
// wavefront needs -- jp = 0: 0, 1, 2; jp = 1: 1, 2, 3 -- and they are fetched as one 8-byte and one 4-byte LDS read whose OFFSETS
// depend on jp, so that the code is the same for both: X, Y = columns (0, 1) | (2, 3), Z = column 2 | 1, V(jj = 0) = X - Z,
// V(jj = 1) = Z + beta Y with beta = +1 | -1.
// The output transform needs all four rows and both column pairs of a tile: two rounds (one per 32-channel block) through the
// 128 KB the patch stages no longer need; wavefront q finishes tile block q & 1, channel group q >> 1 of the round.
template <bool POOL, int ABL = 0, int TUNE = 0>
__global__ void __launch_bounds__(512, 2) wino_bf16x3_p8_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ upk, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncg, int nks, int act, int chunk)
{
    __shared__ __attribute__((aligned(16))) float lds[32768];
    // workgroup -> (spatial block, 64-channel output group): every XCD walks its own contiguous raster range of Sx spatial blocks in
    // chunks of `chunk` blocks, and inside a chunk all blocks of one output group before the next group.  chunk = 1 is "output groups
    // innermost" (the groups of a block run back to back and share its patches in L2) -- right while the packed filters of ALL groups
    // fit the XCD's 4 MB L2 beside them; for the 196- / 256-channel layers they do not (5 - 6 MB: every fragment request missed L2,
    // 32 GB of fabric reads per launch against 4.9 GB algorithmic, profiles/r04_pmc_loftr_l1out2.json), so there a chunk is 8
    // blocks: one group's fragments (1.2 - 1.5 MB) stay L2-resident for 8 workgroups and the chunk's patches (~2 MB) for all groups.
    const int id = blockIdx.x;
    const int xcd = id & 7, jq = id >> 3;
    const int per = chunk * ncg;
    const int ck = jq / per, rr = jq - ck * per;
    const int left = min(chunk, Sx - ck * chunk);           // blocks in this (possibly last, partial) chunk
    if (left <= 0) return;
    const int cg = rr / left, sl = ck * chunk + (rr - cg * left);
    if (cg >= ncg) return;
    const int s = xcd * Sx + sl;
    if (sl >= Sx || s >= S) return;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w >> 1, jp = w & 1;
    const int col = lane & 15, tysub = (lane >> 4) & 1, kg = lane >> 5;
    const int HW = H * W;

    // ---- patch staging by LDS-DMA (see the kernel above): this wavefront stages input channels 2 w, 2 w + 1 of the step
    const int prow = lane / 12, pk = lane - 12 * prow;
    const int pix = 32 * bx - 1 + 4 * pk;
    unsigned voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int iy = 8 * by - 1 + 5 * h + prow;
        const bool ok = prow < 5 && pk < 9 && iy >= 0 && iy < H && pix < W && pix + 3 >= 0;
        voff[h] = ok ? (unsigned)(iy * W + max(pix, 0)) * 4u : WB_OOB;
    }
    const bool fixl = pix < 0;
    unsigned keep = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) if (pix + d >= 0 && pix + d < W) keep |= 1u << d;
    const bool edge = (bx == 0) || (32 * bx + 35 >= W);
    typedef unsigned wb_u32x4 __attribute__((ext_vector_type(4)));
    wb_u32x4 xdesc;
    {
        const unsigned long long xa = (unsigned long long)(x + (size_t)b * Cin * HW);
        xdesc.x = __builtin_amdgcn_readfirstlane((unsigned)xa);
        xdesc.y = __builtin_amdgcn_readfirstlane((unsigned)(xa >> 32) & 0xffffu);
        xdesc.z = (unsigned)(Cin * HW) * 4u;
        xdesc.w = WB_RSRC_FLAGS;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    auto pdma = [&](int c, int buf) {
        if (ABL & (8 | 16)) return;
        if (prow < 5) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned so = (unsigned)(16 * c + 2 * w + q) * (unsigned)HW * 4u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned m0v = lds0 + 4u * (unsigned)(buf * WB_STAGE + (2 * w + q) * WB_CH + (5 * h) * WB_RS);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(voff[h]), "s"(xdesc), "s"(so) : "memory");
                }
            }
        }
    };
    auto pfix = [&](int buf) {
        if (edge && prow < 5 && (fixl || keep != 0xfu)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint4 *pp = (uint4 *)(lds + buf * WB_STAGE + (2 * w + (k >> 1)) * WB_CH + (5 * (k & 1)) * WB_RS + lane * 4);
                uint4 v = *pp;
                if (fixl) v = make_uint4(0u, v.x, v.y, v.z);
                v.x = (keep & 1u) ? v.x : 0u; v.y = (keep & 2u) ? v.y : 0u; v.z = (keep & 4u) ? v.z : 0u; v.w = (keep & 8u) ? v.w : 0u;
                *pp = v;
            }
        }
    };

    // ---- row combination of Winograd row wi (as above) for the three patch columns of column pair jp
    const int ra = (wi == 0) ? 0 : (wi == 2) ? 2 : 1;
    const int rb = (wi == 0) ? 2 : (wi == 2) ? 1 : (wi == 1) ? 2 : 3;
    const float sg = (wi == 1) ? 1.0f : -1.0f;
    const float beta = jp ? -1.0f : 1.0f;
    const int cX = jp ? 2 : 0, cZ = jp ? 1 : 2;
    const int ofa = (2 * tysub + ra) * WB_RS + 2 * col, ofb = (2 * tysub + rb) * WB_RS + 2 * col;    // tile block nb: + 4 nb WB_RS
    float wX[8], wY[8], wZ[8];
    auto wread = [&](int buf, int nb) {
        const wb_lds_f32 *st = (const wb_lds_f32 *)lds + buf * WB_STAGE + (8 * kg) * WB_CH + 4 * nb * WB_RS;
#pragma unroll
        for (int e0 = 0; e0 < 8; e0 += 4) {
            wb_f32x2 a2[4], b2[4];
            float a1[4], b1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const wb_lds_f32 *ch = st + (e0 + e) * WB_CH;
                if (ABL & (8 | 32)) { a2[e] = wb_f32x2{1.f + e0, 2.f + nb}; b2[e] = wb_f32x2{3.f + e, 4.f + lane}; a1[e] = 5.f + e; b1[e] = 6.f + lane; continue; }
                a2[e] = *(const volatile wb_lds_f32x2 *)(ch + ofa + cX); a1[e] = *(const volatile wb_lds_f32 *)(ch + ofa + cZ);
                b2[e] = *(const volatile wb_lds_f32x2 *)(ch + ofb + cX); b1[e] = *(const volatile wb_lds_f32 *)(ch + ofb + cZ);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wX[e0 + e] = __builtin_fmaf(sg, b2[e].x, a2[e].x); wY[e0 + e] = __builtin_fmaf(sg, b2[e].y, a2[e].y);
                wZ[e0 + e] = __builtin_fmaf(sg, b1[e], a1[e]);
            }
            // pin the combinations here (otherwise they sink below the next batch's reads and the raw values pile up)
            asm volatile("" : "+v"(wX[e0]), "+v"(wY[e0]), "+v"(wZ[e0]), "+v"(wX[e0 + 1]), "+v"(wY[e0 + 1]), "+v"(wZ[e0 + 1]),
                              "+v"(wX[e0 + 2]), "+v"(wY[e0 + 2]), "+v"(wZ[e0 + 2]), "+v"(wX[e0 + 3]), "+v"(wY[e0 + 3]), "+v"(wZ[e0 + 3]) :: "memory");
        }
    };
    // V(wi, 2 jp + jj) of the tile block, split and packed two input channels at a time
    auto vmake = [&](WbFrag (&vf)[3], int jj) {
        if (ABL & 4) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) vf[t].u[k] = __float_as_uint(wX[2 * k]) ^ __float_as_uint(jj ? wY[2 * k + 1] : wZ[2 * k + 1]) ^ (unsigned)t;
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h[2], m[2], l[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = 2 * k + e;
                const float v = jj ? __builtin_fmaf(beta, wY[q], wZ[q]) : wX[q] - wZ[q];
                wb_split3(v, h[e], m[e], l[e]);
            }
            vf[0].u[k] = wb_pack(h[0], h[1]);
            vf[1].u[k] = wb_pack(m[0], m[1]);
            vf[2].u[k] = wb_pack(l[0], l[1]);
        }
    };

    // ---- filter fragments of positions (wi, 2 jp), (wi, 2 jp + 1): [jj][mb][term], 12 consecutive fragments of the packed layout
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)upk, 0, (int)(ncg * nks * WB_FRAGS_PER_KSTEP * 1024), WB_RSRC_FLAGS);
    const unsigned fbase = (unsigned)((cg * nks * WB_FRAGS_PER_KSTEP + wi * 24 + jp * 12) * 1024);
    const unsigned lane16 = (unsigned)lane * 16u;
    WbFrag F[2][2][3];
    auto aload = [&](int c, int jj) {
        const unsigned so = fbase + (unsigned)((c * WB_FRAGS_PER_KSTEP + jj * 6) * 1024);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (ABL & 2) F[jj][mb][t].q = make_uint4(0x3f803f80u + lane, 0x3f803f80u + jj, 0x3f803f80u + t, 0x3f803f80u + c);
                else F[jj][mb][t].q = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsF, lane16, so + (unsigned)((mb * 3 + t) * 1024), 0));
            }
    };
#define WB8_PROD(jj, nb, vf, ta, tb) do { \
        if (ABL & 64) { acc[jj][0][nb][ta] += __uint_as_float(F[jj][0][ta].u[tb] ^ vf[tb].u[ta]); acc[jj][1][nb][ta] += __uint_as_float(F[jj][1][ta].u[tb] ^ vf[tb].u[ta]); } \
        else { acc[jj][0][nb] = WB_MFMA(F[jj][0][ta].v, vf[tb].v, acc[jj][0][nb]); acc[jj][1][nb] = WB_MFMA(F[jj][1][ta].v, vf[tb].v, acc[jj][1][nb]); } } while (0)
#define WB8_PHASE(jj, nb, vf) do { WB8_PROD(jj, nb, vf, 1, 1); WB8_PROD(jj, nb, vf, 0, 2); WB8_PROD(jj, nb, vf, 2, 0); \
                                   WB8_PROD(jj, nb, vf, 0, 1); WB8_PROD(jj, nb, vf, 1, 0); WB8_PROD(jj, nb, vf, 0, 0); } while (0)
#define WB8_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))      /* vmcnt(n) only */

    // ---- prologue
    aload(0, 0); aload(0, 1);
    pdma(0, 0);
    WB8_VMCNT(0);
    pfix(0);
    __syncthreads();
    f32x16 acc[2][2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jj][mb][nb][r] = 0.f;

    // ---- K loop, software-pipelined.  A wavefront that alternates "produce V (52 VALU)" and "12 MFMAs" is blocked at the MFMA
    // issue while the matrix pipe drains -- and with the barrier its SIMD partner is in the same phase at the same time, so
    // nothing overlaps (6.5 ms: the sum).  tools/ubench/mfma_valu_bf16.hip: five VALU instructions per MFMA are free when they
    // sit BETWEEN the MFMAs in program order.  So V is double-buffered (vfA / vfB) and every block of 12 MFMAs is written
    // together with the production of the NEXT block's V:
    //     S1  products (jj 0, nb 0; vfA)   +  V(jj 1, nb 0) -> vfB
    //     S2  products (jj 1, nb 0; vfB)   +  row combinations of tile block 1, V(jj 0, nb 1) -> vfA
    //     S3  products (jj 0, nb 1; vfA)   +  V(jj 1, nb 1) -> vfB;   fragments (jj 0) of the next step requested
    //     -- vmcnt: the next step's patches are in; border fix-up; barrier --
    //     S4  products (jj 1, nb 1; vfB)   +  row combinations of tile block 0 of the NEXT step, V(jj 0, nb 0) -> vfA;
    //         fragments (jj 1) of the next step requested
    // Top of a step: vmcnt(0) (the step's fragments are in; hipcc would otherwise wait vmcnt(0) at their first use, i.e. for
    // the DMAs it cannot see), then the patch DMAs of the step after go out: three blocks of time to land.
    // TUNE & 1: pin the interleave -- one MFMA, then nv VALU (+ nd LDS reads) -- with sched_group_barrier instead of leaving it to hipcc
#define WB8_SCHED(nv, nd) do { if (TUNE & 1) { _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); \
        if (nd) __builtin_amdgcn_sched_group_barrier(0x100, nd, 0); __builtin_amdgcn_sched_group_barrier(0x2, nv, 0); } } } while (0)
    WbFrag vfA[3], vfB[3];
    wread(0, 0);
    vmake(vfA, 0);
    int c = 0;
    for (; c + 1 < nks; ++c) {
        WB8_VMCNT(0);
        pdma(c + 1, (c + 1) & 1);
        vmake(vfB, 1);
        WB8_PHASE(0, 0, vfA);
        WB8_SCHED(5, 0);
        if (TUNE & 2) __builtin_amdgcn_sched_barrier(0);
        wread(c & 1, 1);
        vmake(vfA, 0);
        WB8_PHASE(1, 0, vfB);
        WB8_SCHED(7, 3);
        if (TUNE & 2) __builtin_amdgcn_sched_barrier(0);
        vmake(vfB, 1);
        WB8_PHASE(0, 1, vfA);
        WB8_SCHED(5, 0);
        aload(c + 1, 0);
        WB8_VMCNT(6);                                                     // the four DMAs are older than the six fragment loads
        pfix((c + 1) & 1);
        __syncthreads();
        wread((c + 1) & 1, 0);
        vmake(vfA, 0);
        WB8_PHASE(1, 1, vfB);
        WB8_SCHED(7, 3);
        aload(c + 1, 1);
    }
    WB8_VMCNT(0);
    vmake(vfB, 1);
    WB8_PHASE(0, 0, vfA);
    WB8_SCHED(5, 0);
    wread(c & 1, 1);
    vmake(vfA, 0);
    WB8_PHASE(1, 0, vfB);
    WB8_SCHED(7, 3);
    vmake(vfB, 1);
    WB8_PHASE(0, 1, vfA);
    WB8_SCHED(5, 0);
    WB8_PHASE(1, 1, vfB);
#undef WB8_SCHED
#undef WB8_PHASE
#undef WB8_PROD
#undef WB8_VMCNT

    if (ABL & 1) {
        float t = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[jj][mb][nb][r];
        if (t == 12345.678f) y[tid] = t;
        return;
    }
    // ---- output transform.  Row partials over j: pa = M0 + M1 + M2, pb = M1 - M2 - M3; wavefront (wi, 0) contributes (M0 + M1, M1),
    // wavefront (wi, 1) contributes (M2, -(M2 + M3)).  Round mb: part[row][jp][ab][nb][r4][lane] (float4 = registers 4 r4 .. 4 r4 + 3),
    // 8 x 16 KB; wavefront q then finishes tile block q & 1, register group q >> 1 (four channels) of the round.
    const int qnb = w & 1, qr4 = w >> 1;
    const int tr = 2 * qnb + tysub;
    const int ty = 4 * by + tr, tx = 16 * bx + col;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const size_t cstride = (size_t)Ho * Wo;
    float4 *part = (float4 *)lds;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        __syncthreads();                                    // patch stages (mb = 0) / the previous round's partials (mb = 1) are dead
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * r4 + k;
                    const float sum = acc[0][mb][nb][r] + acc[1][mb][nb][r];
                    pa[k] = jp ? acc[0][mb][nb][r] : sum;
                    pb[k] = jp ? -sum : acc[1][mb][nb][r];
                }
                part[((((wi * 2 + jp) * 2 + 0) * 2 + nb) * 4 + r4) * 64 + lane] = make_float4(pa[0], pa[1], pa[2], pa[3]);
                part[((((wi * 2 + jp) * 2 + 1) * 2 + nb) * 4 + r4) * 64 + lane] = make_float4(pb[0], pb[1], pb[2], pb[3]);
            }
        const int co0 = cg * 64 + mb * 32 + 4 * kg + 8 * qr4;       // this lane's four channels of the round: co0 + k
        float bv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) bv[k] = bias ? bias[min(co0 + k, Cout - 1)] : 0.f;
        __syncthreads();
        const float4 *pq = part + (qnb * 4 + qr4) * 64 + lane;      // + ((row * 2 + jp) * 2 + ab) * 8 * 64
        float4 P[4][2];
#pragma unroll
        for (int row = 0; row < 4; ++row)
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) {
                const float4 u = pq[(((row * 2 + 0) * 2 + ab) * 8) * 64], v = pq[(((row * 2 + 1) * 2 + ab) * 8) * 64];
                P[row][ab] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
            }
        float Y[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#define WB_EL(v) (k == 0 ? (v).x : k == 1 ? (v).y : k == 2 ? (v).z : (v).w)
            Y[k][0] = ((WB_EL(P[0][0]) + WB_EL(P[1][0])) + WB_EL(P[2][0])) + bv[k];
            Y[k][1] = ((WB_EL(P[0][1]) + WB_EL(P[1][1])) + WB_EL(P[2][1])) + bv[k];
            Y[k][2] = ((WB_EL(P[1][0]) - WB_EL(P[2][0])) - WB_EL(P[3][0])) + bv[k];
            Y[k][3] = ((WB_EL(P[1][1]) - WB_EL(P[2][1])) - WB_EL(P[3][1])) + bv[k];
#undef WB_EL
        }
        float *yb = y + ((size_t)b * Cout + co0) * cstride;
        const bool allco = cg * 64 + mb * 32 + 32 <= Cout;
        if (POOL) {
            float m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = fmaxf(fmaxf(Y[k][0], Y[k][1]), fmaxf(Y[k][2], Y[k][3]));
                if (act == 1) m[k] = fmaxf(m[k], 0.f);
                else if (act == 2) m[k] = m[k] > 0.f ? m[k] : 0.01f * m[k];
            }
            if (ty < Ho && tx < Wo) {
                float *yo = yb + (size_t)ty * Wo + tx;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (allco || co0 + k < Cout) yo[(size_t)k * cstride] = m[k];
            }
        } else {
            const int oy = 2 * ty, ox = 2 * tx;
            const bool c0 = ox < W, c1 = ox + 1 < W, r0 = oy < H, r1 = oy + 1 < H;
            if (residual) {
                const float *rb0 = residual + ((size_t)b * Cout + co0) * cstride + (size_t)oy * W + ox;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (allco || co0 + k < Cout) {
                        const float *ro = rb0 + (size_t)k * cstride;
                        if (r0 && c0) Y[k][0] += ro[0];
                        if (r0 && c1) Y[k][1] += ro[1];
                        if (r1 && c0) Y[k][2] += ro[W];
                        if (r1 && c1) Y[k][3] += ro[W + 1];
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (act == 1) Y[k][q] = fmaxf(Y[k][q], 0.f);
                    else if (act == 2) Y[k][q] = Y[k][q] > 0.f ? Y[k][q] : 0.01f * Y[k][q];
                }
            float *yo0 = yb + (size_t)oy * W + ox;
            const bool interior = allco && r0 && r1 && c1 && !(W & 1);
            if (interior) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float *yo = yo0 + (size_t)k * cstride;
                    *(float2 *)yo = make_float2(Y[k][0], Y[k][1]);
                    *(float2 *)(yo + W) = make_float2(Y[k][2], Y[k][3]);
                }
            } else
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(allco || co0 + k < Cout)) continue;
                float *yo = yo0 + (size_t)k * cstride;
                if (c1) {
                    if (!(W & 1)) {
                        if (r0) *(float2 *)yo = make_float2(Y[k][0], Y[k][1]);
                        if (r1) *(float2 *)(yo + W) = make_float2(Y[k][2], Y[k][3]);
                    } else {
                        if (r0) { yo[0] = Y[k][0]; yo[1] = Y[k][1]; }
                        if (r1) { yo[W] = Y[k][2]; yo[W + 1] = Y[k][3]; }
                    }
                } else if (c0) {
                    if (r0) yo[0] = Y[k][0];
                    if (r1) yo[W] = Y[k][2];
                }
            }
        }
    }
}

extern "C" {

size_t mfr_wino_bf16x3_filter_bytes(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    const size_t ncg = (Cout + 63) / 64, nks = (Cin + 15) / 16;
    return ncg * nks * WB_FRAGS_PER_KSTEP * 1024;
}

int mfr_wino_bf16x3_filter_transform(const float *w, int Cin, int Cout, void *upk, void *stream)
{
    if (!w || !upk || Cin <= 0 || Cout <= 0) return MFR_E_ARG;
    const long long total = (long long)(mfr_wino_bf16x3_filter_bytes(Cin, Cout) / 16);
    hipLaunchKernelGGL(wb_filter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, (Cin + 15) / 16, total,
                       (uint4 *)upk);
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv3x3_wino_bf16x3_variant(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                                    int act, int pool, int variant, float *y, void *stream)
{
    if (!x || !upk || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || act < 0 || act > 2) return MFR_E_ARG;
    if (pool && (H < 2 || W < 2 || residual)) return MFR_E_ARG;
    if ((size_t)4 * Cin * H * W >= 0x7fffffffull) return MFR_E_ARG;    // one image must fit a 2 GB buffer descriptor
    const int nbx = ((W + 1) / 2 + 15) / 16, nby = ((H + 1) / 2 + 3) / 4;
    const int ncg = (Cout + 63) / 64, nks = (Cin + 15) / 16;
    const long long S = (long long)nbx * nby * B, Sx = (S + 7) / 8, grid = Sx * 8 * ncg;
    if (grid > 0x7fffffffll) return MFR_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define WB_GO(K) hipLaunchKernelGGL(K, dim3((unsigned)grid), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg, nks, act)
    if (variant == 0 || variant == 3 || (variant >= 400 && variant < 528 && pool)) {     // default since round 4 (501..503: TUNE builds, results right)     // eight wavefronts per workgroup (400 + ABL: timing ablations, results are wrong)
        const int chunk = ((size_t)ncg * nks * WB_FRAGS_PER_KSTEP * 1024 > (size_t)(2u << 20)) ? 8 : 1;       // packed filters of all groups vs half an XCD's L2
        const long long grid8 = ((Sx + chunk - 1) / chunk) * (long long)chunk * ncg * 8;
        if (grid8 > 0x7fffffffll) return MFR_E_ARG;
#define WB_GO8(K) hipLaunchKernelGGL(K, dim3((unsigned)grid8), dim3(512), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg, nks, act, chunk)
        if (variant == 0 || variant == 3) { if (pool) WB_GO8((wino_bf16x3_p8_kernel<true>)); else WB_GO8((wino_bf16x3_p8_kernel<false>)); }
        else if (variant == 501 && pool) WB_GO8((wino_bf16x3_p8_kernel<true, 0, 1>));
        else if (variant == 502 && pool) WB_GO8((wino_bf16x3_p8_kernel<true, 0, 2>));
        else if (variant == 503 && pool) WB_GO8((wino_bf16x3_p8_kernel<true, 0, 3>));
        else switch (variant - 400) {
#define WB_ABL8(A) case A: WB_GO8((wino_bf16x3_p8_kernel<true, A>)); break;
        WB_ABL8(1) WB_ABL8(2) WB_ABL8(4) WB_ABL8(8) WB_ABL8(12) WB_ABL8(15) WB_ABL8(64) WB_ABL8(16) WB_ABL8(32) WB_ABL8(71) WB_ABL8(79) WB_ABL8(103) WB_ABL8(87)
#undef WB_ABL8
        default: return MFR_E_ARG;
        }
#undef WB_GO8
    }
    else if (variant >= 200 && variant < 328 && pool) {     // timing ablations of the two-workgroups-per-CU kernel: 200 + ABL (results are wrong)
        const int ncg2 = (Cout + 31) / 32;
        const long long grid2 = Sx * 8 * ncg2;
        switch (variant - 200) {
#define WB_ABL2(A) case A: hipLaunchKernelGGL((wino_bf16x3_w2_kernel<true, A>), dim3((unsigned)grid2), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg2, nks, act); break;
        WB_ABL2(1) WB_ABL2(2) WB_ABL2(4) WB_ABL2(8) WB_ABL2(12) WB_ABL2(14) WB_ABL2(15) WB_ABL2(64) WB_ABL2(16) WB_ABL2(32)
        case 101: hipLaunchKernelGGL((wino_bf16x3_w2_kernel<true, 0, 1>), dim3((unsigned)grid2), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg2, nks, act); break;
        case 102: hipLaunchKernelGGL((wino_bf16x3_w2_kernel<true, 0, 2>), dim3((unsigned)grid2), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg2, nks, act); break;
        case 103: hipLaunchKernelGGL((wino_bf16x3_w2_kernel<true, 0, 3>), dim3((unsigned)grid2), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg2, nks, act); break;
#undef WB_ABL2
        default: return MFR_E_ARG;
        }
    }
    else if (variant == 2) {                           // two workgroups per CU: 64 tiles x 32 output channels each
        const int ncg2 = (Cout + 31) / 32;
        const long long grid2 = Sx * 8 * ncg2;
        if (grid2 > 0x7fffffffll) return MFR_E_ARG;
        if (pool) hipLaunchKernelGGL((wino_bf16x3_w2_kernel<true>), dim3((unsigned)grid2), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg2, nks, act);
        else hipLaunchKernelGGL((wino_bf16x3_w2_kernel<false>), dim3((unsigned)grid2), dim3(256), 0, st, x, (const uint4 *)upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg2, nks, act);
    }
    else if (variant == 32) { if (pool) WB_GO((wino_bf16x3_kernel<true>)); else WB_GO((wino_bf16x3_kernel<false>)); }
    else if (pool) {                                   // timing ablations (tools/bench_conv.py): pooled layers only
        switch (variant) {
#define WB_ABL(A) case A: WB_GO((wino_bf16x3_kernel<true, A>)); break;
        WB_ABL(1) WB_ABL(2) WB_ABL(4) WB_ABL(8) WB_ABL(3) WB_ABL(7) WB_ABL(15) WB_ABL(14) WB_ABL(6) WB_ABL(12) WB_ABL(16)
#undef WB_ABL
        default: return MFR_E_ARG;
        }
    } else return MFR_E_ARG;
#undef WB_GO
    CHECK_LAUNCH();
    return 0;
}

/* debug: the s_memtime stamps of the last variant-16 launch (4 wavefronts x 64 stamps) */
int mfr_wino_bf16x3_profile(unsigned long long *out_host)
{
    if (!out_host) return MFR_E_ARG;
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(wb_prof), sizeof(unsigned long long) * 256, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : MFR_E_LAUNCH;
}

int mfr_conv3x3_wino_bf16x3(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                            int act, int pool, float *y, void *stream)
{
    return mfr_conv3x3_wino_bf16x3_variant(x, upk, bias, residual, B, Cin, Cout, H, W, act, pool, 0, y, stream);
}

}  // extern "C"
