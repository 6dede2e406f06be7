// winograd_conv.hip -- 3x3 / stride 1 / pad 1 convolution (NCHW f32) of the SuperPoint encoder as a fused
// Winograd F(2x2, 3x3) kernel on the gfx950 matrix cores, with the layer epilogue (+bias, ReLU, optional
// 2x2 max-pool) folded into the output transform.
//
// Reference call site: SuperGlue_matcher (etc/feature_matching_baselines/matchers.py:62-120) -> upstream
// SuperPoint encoder conv1b..conv4b, convPa, convDa (un-vendored; restated in SURVEY.md Appendix A.2):
// 8 of the 12 convolutions, >95 % of the encoder FLOPs (conv1b alone: 0.92 TFLOP per 16-pair step).
// The library path (MIOpen's fp32 Winograd on the VALU) runs them at ~107 TFLOP/s direct-equivalent and
// was half of the whole step; F(2,3) needs 2.25x fewer multiplications, and here they run on the fp32
// MFMA pipe (v_mfma_f32_16x16x4_f32, 157 TFLOP/s dense).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d = 4x4 input patch, Y = 2x2 outputs, per (cin, cout)
//   => 16 independent GEMMs  M_xi[cout, tile] = sum_cin U_xi[cout, cin] V_xi[cin, tile]
//
// Mapping to CDNA4:
//   * workgroup = 4 wavefronts = 4 tile rows x 16 tiles (8 x 32 output pixels) x 16*NBLK output channels;
//     wavefront w owns tile row w: 16 tiles = the N dimension of the 16x16x4 MFMA, NBLK M-blocks of 16 couts,
//     all 16 Winograd positions xi -> 16 x NBLK accumulators of 4 registers.  NBLK = 2: 128 accumulator
//     registers, two workgroups per CU; NBLK = 4: 256 accumulator registers (the whole AGPR file), one
//     wavefront per SIMD -- the input patch loads and the input transform are amortised over twice the MFMAs
//     (the NBLK = 2 stream is issue-bound: ~3 non-MFMA instructions per MFMA with two waves per SIMD).
//   * the K dimension is cin, 4 per MFMA.  Lane (kq = lane>>4, col = lane&15) loads the 4x4 input patch of
//     (cin = 4c + kq, tile col), transforms it in registers (32 adds) -- and the 16 results ARE that lane's
//     B operands for the 16 xi MFMAs: the input transform never touches LDS.
//   * U (transformed filters) is pre-packed once per weight set in exactly the order the lanes consume it;
//     a 4*NBLK KB slab per (cin chunk, cout block) is staged through LDS (double buffered, one barrier per
//     chunk) and read back as conflict-free ds_read_b128: one read = A operands of 4 MFMAs.
//   * input patches come through buffer loads: zero padding = out-of-range offsets (hardware returns 0); each
//     lane loads its two own columns as one dwordx2 and takes the other two from its neighbours' registers
//     with DPP row shifts (16-lane row = one tile row); only the edge lanes fetch a halo element.
//   * all 16 xi of a (tile, cout) end in the same lane, so the output transform, bias, ReLU and the 2x2
//     max-pool (= exactly one Winograd tile) run in registers; the activation is written once.
//   * grid: 1-D, remapped so that every XCD walks one contiguous raster range of spatial blocks (cout blocks
//     innermost): halo rows/columns and the second cout block are L2 hits.
// Per cin chunk and wavefront: 8 buffer loads, 8 DPP moves + 32 adds, 4*NBLK ds_read_b128, 16*NBLK MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "zero_fill.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef WN_SCHED_ALL
#define WN_SCHED_ALL 1         // transform completes before the MFMA run of the chunk (fewer register copies; measured +1.5 %)
#endif
#define WN_TX 16                 // tiles along x per workgroup
#define WN_TY 4                  // tile rows per workgroup (= wavefronts)

// dword3 of a raw buffer descriptor on gfx9-family CDNA (32-bit data format); out-of-range reads return 0,
// which is how the zero padding of the convolution is produced: padded taps get an offset beyond the buffer
#define WN_RSRC_FLAGS 0x00020000
#define WN_OOB 0x80000000u

// NBLK (16-cout MFMA blocks per workgroup) is 2: 128 accumulator registers, two workgroups per CU.  (A one-wavefront-per-SIMD
// NBLK = 4 instantiation was measured in round 1 -- 14.3 vs 12.2 ms over the nine SuperPoint layers, its un-overlapped
// prologue / epilogue cost more than the halved transform work saved -- and is no longer built.)  The packed-filter layout
// depends on NBLK only, so it is a compile-time constant of the library: no environment variable, no hidden state.
#define WN_NBLK 2
static inline int wn_nblk(int) { return WN_NBLK; }

// packed U_xi[cout][cin]: [chunk c = cin/4][cout block cb][q = xi/4][blk][lane = (cin%4)*16 + cout%16][e = xi%4]
__global__ void __launch_bounds__(256) wino_filter_kernel(const float *__restrict__ w, int Cin, int Cout, int CoutP, int nblk, float *__restrict__ upk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin * Cout) return;
    const int co = i / Cin, ci = i - co * Cin;
    const float *g = w + (size_t)i * 9;
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
        t[0][j] = g0; t[1][j] = 0.5f * (g0 + g1 + g2); t[2][j] = 0.5f * (g0 - g1 + g2); t[3][j] = g2;
    }
    const int wco = 16 * nblk, ncb = CoutP / wco, slab = 1024 * nblk;   // couts >= Cout stay zero (buffer pre-cleared)
    const int c = ci >> 2, kq = ci & 3, cb = co / wco, blk = (co % wco) >> 4, row = co & 15;
    float *dst = upk + ((size_t)c * ncb + cb) * slab;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float u[4];
        u[0] = t[r][0]; u[1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]); u[2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]); u[3] = t[r][2];
        if (r == 3) { u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; u[3] = -u[3]; }   // row 3 is carried negated on both operands (see wino_conv3x3_shared_kernel)
        // xi = 4 r + j  ->  q = r, e = j
        *(float4 *)(dst + (((size_t)r * nblk + blk) * 64 + kq * 16 + row) * 4) = make_float4(u[0], u[1], u[2], u[3]);
    }
}

// output transform + bias + activation (+ 2x2 max-pool) + store, shared by the kernel variants
// BIAS = false: the caller folded the bias into an accumulator; ACT >= 0: the activation is a compile-time constant (every VALU
// instruction counts next to the f32 MFMAs, also the selects of a run-time activation code)
template <bool POOL, int LOAD, int NBLK, bool BIAS = true, int ACT = -1>
__device__ __forceinline__ void wn_epilogue(f32x4 (&acc)[16][NBLK], const float *__restrict__ bias, float *__restrict__ y,
                                            const float *__restrict__ residual, int Cout, int H, int W, int act_rt, int b, int cb, int kq, int ty, int tx)
{
    const int act = ACT >= 0 ? ACT : act_rt;
    constexpr int WCO = 16 * NBLK;
    // output transform A^T M A + bias (+ReLU) (+2x2 max-pool) in registers; accumulator register r of
    // M-block blk = cout cb*WCO + blk*16 + 4 kq + r, column = this lane's tile
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cb * WCO + blk * 16 + 4 * kq + r;
            float a0[4], a1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[j] = acc[j][blk][r] + acc[4 + j][blk][r] + acc[8 + j][blk][r];
                a1[j] = acc[4 + j][blk][r] - acc[8 + j][blk][r] - acc[12 + j][blk][r];
            }
            if (co >= Cout) continue;                        // padded output channels (Cout not a multiple of 32)
            float y00 = a0[0] + a0[1] + a0[2], y01 = a0[1] - a0[2] - a0[3];
            float y10 = a1[0] + a1[1] + a1[2], y11 = a1[1] - a1[2] - a1[3];
            if (BIAS) {
                const float bv = bias ? bias[co] : 0.f;
                y00 += bv; y01 += bv; y10 += bv; y11 += bv;
            }
            const size_t plane = ((size_t)b * Cout + co) * ((size_t)Ho * Wo);
            float *yo = y + plane;
            if (POOL) {
                float m = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));   // act is monotone: act(max) = max(act)
                m = (act == 1) ? fmaxf(m, 0.f) : (act == 2) ? (m > 0.f ? m : 0.01f * m) : m;
                if (ty < Ho && tx < Wo) yo[(size_t)ty * Wo + tx] = m;
            } else {
                const int oy = 2 * ty, ox = 2 * tx;
                if (residual) {
                    const float *ro = residual + plane;
                    if (oy < H && ox < W) y00 += ro[(size_t)oy * W + ox];
                    if (oy < H && ox + 1 < W) y01 += ro[(size_t)oy * W + ox + 1];
                    if (oy + 1 < H && ox < W) y10 += ro[(size_t)(oy + 1) * W + ox];
                    if (oy + 1 < H && ox + 1 < W) y11 += ro[(size_t)(oy + 1) * W + ox + 1];
                }
                if (act == 1) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
                else if (act == 2) {
                    y00 = y00 > 0.f ? y00 : 0.01f * y00; y01 = y01 > 0.f ? y01 : 0.01f * y01;
                    y10 = y10 > 0.f ? y10 : 0.01f * y10; y11 = y11 > 0.f ? y11 : 0.01f * y11;
                }
                if (ox + 1 < W) {
                    if (LOAD == 1) {                         // W even: 8-byte aligned pairs
                        if (oy < H) *(float2 *)(yo + (size_t)oy * W + ox) = make_float2(y00, y01);
                        if (oy + 1 < H) *(float2 *)(yo + (size_t)(oy + 1) * W + ox) = make_float2(y10, y11);
                    } else {
                        if (oy < H) { yo[(size_t)oy * W + ox] = y00; yo[(size_t)oy * W + ox + 1] = y01; }
                        if (oy + 1 < H) { yo[(size_t)(oy + 1) * W + ox] = y10; yo[(size_t)(oy + 1) * W + ox + 1] = y11; }
                    }
                } else if (ox < W) {
                    if (oy < H) yo[(size_t)oy * W + ox] = y00;
                    if (oy + 1 < H) yo[(size_t)(oy + 1) * W + ox] = y10;
                }
            }
        }
    }
}

// LOAD: 0 = 16 dword loads per patch (any layout); 1 = dwordx2 + neighbour sharing, W even; 2 = same, W odd (the
// last column's pair partner belongs to the next row and is masked)
template <bool POOL, int LOAD, int NBLK>
__global__ void __launch_bounds__(256, NBLK == 2 ? 2 : 1) wino_conv3x3_kernel(
    const float *__restrict__ x, const float *__restrict__ upk, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncb, int act)
{
    constexpr int SLAB = 1024 * NBLK;                        // floats of packed U per (cin chunk of 4, cout block)
    constexpr bool PAIR = LOAD != 0;
    constexpr bool SCHED = WN_SCHED_ALL || NBLK == 4;
    __shared__ __attribute__((aligned(16))) float Us[2][SLAB];
    // workgroup id -> (XCD, position in that XCD's queue): ids are dealt round-robin to the 8 XCDs, so XCD x runs ids
    // x, x+8, x+16, ...  Each XCD gets one CONTIGUOUS raster range of spatial blocks, cout blocks innermost: the
    // cout blocks of a spatial block and its row/column neighbours (which share the 2-pixel halo) then run back to back
    // on the same L2 instead of on 8 different ones (PMC: HBM-side reads 2.3x the input with the interleaved map).
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int cb = jj % ncb;
    const int sl = jj / ncb;
    const int s = xcd * Sx + sl;
    if (sl >= Sx || s >= S) return;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int ty = by * WN_TY + w, tx = bx * WN_TX + col;
    const int HW = H * W;

    // one buffer per image: [Cin, H, W] f32; lane byte offsets inside a 4-channel slab, chunk advance in soffset
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)b * Cin * HW), 0, Cin * HW * 4, WN_RSRC_FLAGS);
    constexpr int NOFF = PAIR ? 8 : 16;
    constexpr int NRAW = PAIR ? 12 : 16;
    unsigned off[NOFF];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int iy = 2 * ty - 1 + a;
        const bool rv = iy >= 0 && iy < H;
        const unsigned rowb = (unsigned)(kq * HW + iy * W) * 4u;
        if (PAIR) {
            off[a] = (rv && 2 * tx < W) ? rowb + 8u * tx : WN_OOB;
            const int hx = (col == 0) ? 2 * tx - 1 : 2 * tx + 2;
            off[4 + a] = (rv && (col == 0 || col == 15) && hx >= 0 && hx < W) ? rowb + 4u * hx : WN_OOB;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = 2 * tx - 1 + c;
                off[a * 4 + c] = (rv && ix >= 0 && ix < W) ? rowb + 4u * ix : WN_OOB;
            }
        }
    }
    const bool p1ok = 2 * tx + 1 < W;                        // LOAD == 2 only
    const float4 *ub = (const float4 *)(upk + (size_t)cb * SLAB) + tid;
    const size_t ustride = (size_t)ncb * (SLAB / 4);         // float4 per chunk
    const int nchunks = Cin >> 2;
    const unsigned cstep = 16u * HW;                         // bytes per 4-channel chunk

    auto gload = [&](unsigned (&raw)[NRAW], int c) {
        const unsigned so = (unsigned)c * cstep;
        if (PAIR) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const auto pr = __builtin_amdgcn_raw_buffer_load_b64(rs, off[a], so, 0);
                raw[3 * a] = pr[0]; raw[3 * a + 1] = pr[1];
                raw[3 * a + 2] = __builtin_amdgcn_raw_buffer_load_b32(rs, off[4 + a], so, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, off[i], so, 0);
        }
    };
    // input transform B^T d B in registers: v[4 i + j] is this lane's B operand of Winograd position (i, j)
    auto transform = [&](const unsigned (&raw)[NRAW], float (&v)[16]) {
        float d[16];                                         // 4x4 patch d[4 a + col] of (cin, this lane's tile)
        if (PAIR) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int p0 = (int)raw[3 * a], hv = (int)raw[3 * a + 2];
                const int p1 = (LOAD == 2 && !p1ok) ? 0 : (int)raw[3 * a + 1];
                // column 2tx-1 = left neighbour's second element, column 2tx+2 = right neighbour's first; the edge
                // lanes of the 16-lane row have no neighbour and keep `old` = their halo load
                d[4 * a] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hv, p1, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
                d[4 * a + 1] = __builtin_bit_cast(float, p0);
                d[4 * a + 2] = __builtin_bit_cast(float, p1);
                d[4 * a + 3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hv, p0, 0x101 /* row_shl:1 */, 0xf, 0xf, false));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = __builtin_bit_cast(float, raw[i]);
        }
        float t[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = d[j] - d[8 + j];
            t[4 + j] = d[4 + j] + d[8 + j];
            t[8 + j] = d[8 + j] - d[4 + j];
            t[12 + j] = d[12 + j] - d[4 + j];               // = -(B^T d)[3][j]: the packed filters carry the same sign
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * i] = t[4 * i] - t[4 * i + 2];
            v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
            v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
            v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
        }
    };

    f32x4 acc[16][NBLK];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < NBLK; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NBLK == 4 runs one wavefront per SIMD: nobody else hides the LDS latency, so the A operands of the whole chunk
    // are requested first and the input transform runs underneath them
    auto aload = [&](const float *slab, float4 (&a)[4 * NBLK]) {
        const float4 *us = (const float4 *)slab + lane;
#pragma unroll
        for (int i = 0; i < 4 * NBLK; ++i) a[i] = us[i * 64];
    };
    auto mfma_chunk = [&](const float4 (&a)[4 * NBLK], const float (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < NBLK; ++k) acc[4 * q][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * NBLK + k].x, v[4 * q], acc[4 * q][k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NBLK; ++k) acc[4 * q + 1][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * NBLK + k].y, v[4 * q + 1], acc[4 * q + 1][k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NBLK; ++k) acc[4 * q + 2][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * NBLK + k].z, v[4 * q + 2], acc[4 * q + 2][k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NBLK; ++k) acc[4 * q + 3][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * NBLK + k].w, v[4 * q + 3], acc[4 * q + 3][k], 0, 0, 0);
        }
    };

    float4 u0, u1, u2, u3;                                   // the chunk's slab: NBLK float4 per thread
    auto uload = [&](int c) {
        const float4 *q = ub + (size_t)c * ustride;
        u0 = q[0]; u1 = q[256];
        if (NBLK == 4) { u2 = q[512]; u3 = q[768]; }
    };
    auto ustore = [&](int st) {
        float4 *d0 = (float4 *)Us[st] + tid;
        d0[0] = u0; d0[256] = u1;
        if (NBLK == 4) { d0[512] = u2; d0[768] = u3; }
    };
    unsigned raw[NRAW];
    gload(raw, 0);
    uload(0);
    ustore(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        float v[16];
        float4 a[4 * NBLK];
        aload(Us[c & 1], a);
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
        transform(raw, v);
        if (more) { gload(raw, c + 1); uload(c + 1); }       // in flight during the MFMAs below
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(a, v);
        if (more) ustore((c + 1) & 1);
        __syncthreads();
    }

    wn_epilogue<POOL, LOAD, NBLK>(acc, bias, y, residual, Cout, H, W, act, b, cb, kq, ty, tx);
}

// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined variant (NBLK = 2, Cin % 8 == 0).  Same mapping, operands and arithmetic as the kernel above (the
// results are bit-identical); what changes is the instruction stream of the K loop.  The kernel above runs a chunk as
// [56 VALU of input transform] -> [32 MFMAs back to back]: two wavefronts that share a SIMD fall into lock step (both in
// their MFMA run, then both in their VALU run), the matrix pipe idles during the VALU runs and PMC shows it 54 % busy with
// 60 % of the wave time issue-stalled.  Here the transform of chunk c+1 is issued UNDER the MFMAs of chunk c, one or two
// VALU instructions per MFMA (sched_group_barrier pins the interleave), so every wavefront's stream is balanced at the
// granularity of a single 32-cycle MFMA and any phase relation between the co-resident wavefronts keeps the pipe fed.
// The loop body is branch-free (one scheduling region): prefetches beyond the last chunk are out-of-range buffer loads
// (return 0) and the loop is unrolled by two so the patch / operand registers ping-pong without copies.
// ---------------------------------------------------------------------------------------------------------------------
#define WN_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define WN_M_VALU 0x2
#define WN_M_MFMA 0x8
#define WN_M_VMEM_RD 0x20
#define WN_M_DS_RD 0x100
#define WN_M_DS_WR 0x200

// MODE 0: pinned interleave (sched_group_barrier); MODE 1: the same software pipeline, placement left to the compiler
// ABL (timing ablations, results are WRONG when != 0; tools/tune_wino.py only): 1 = no output transform / stores,
// 2 = no workgroup barrier in the K loop, 4 = no input transform arithmetic, 8 = no patch loads
template <bool POOL, int LOAD, int MODE, int ABL = 0>
__global__ void __launch_bounds__(256, 2) wino_conv3x3_pipe_kernel(
    const float *__restrict__ x, const float *__restrict__ upk, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncb, int act)
{
    constexpr int NBLK = 2;
    constexpr int SLAB = 1024 * NBLK;
    static_assert(LOAD == 1 || LOAD == 2, "paired loads only");
    __shared__ __attribute__((aligned(16))) float Us[2][SLAB];
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int cb = jj % ncb;
    const int sl = jj / ncb;
    const int s = xcd * Sx + sl;
    if (sl >= Sx || s >= S) return;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int ty = by * WN_TY + w, tx = bx * WN_TX + col;
    const int HW = H * W;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)b * Cin * HW), 0, Cin * HW * 4, WN_RSRC_FLAGS);
    // packed filters as a buffer too: chunk prefetches past the end read zeros instead of needing a branch
    const int nchunks = Cin >> 2;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)upk, 0, nchunks * ncb * SLAB * 4, WN_RSRC_FLAGS);
    unsigned off[8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int iy = 2 * ty - 1 + a;
        const bool rv = iy >= 0 && iy < H;
        const unsigned rowb = (unsigned)(kq * HW + iy * W) * 4u;
        off[a] = (rv && 2 * tx < W) ? rowb + 8u * tx : WN_OOB;
        const int hx = (col == 0) ? 2 * tx - 1 : 2 * tx + 2;
        off[4 + a] = (rv && (col == 0 || col == 15) && hx >= 0 && hx < W) ? rowb + 4u * hx : WN_OOB;
    }
    const bool p1ok = 2 * tx + 1 < W;
    const unsigned cstep = 16u * HW;
    const unsigned uoff = (unsigned)(cb * SLAB + tid * 4) * 4u;      // this thread's first float4 of a chunk's slab
    const unsigned ustep = (unsigned)ncb * SLAB * 4u;                // bytes per chunk

    struct Patch { unsigned p0[4], p1[4], h[4]; };                   // 4 rows: own column pair + halo element
    auto gload = [&](Patch &r, int c) {
        const unsigned so = (unsigned)c * cstep;
        if (ABL & 8) {
#pragma unroll
            for (int a = 0; a < 4; ++a) { r.p0[a] = so + a; r.p1[a] = so ^ a; r.h[a] = so; }
            return;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const auto pr = __builtin_amdgcn_raw_buffer_load_b64(rs, off[a], so, 0);
            r.p0[a] = pr[0]; r.p1[a] = pr[1];
            r.h[a] = __builtin_amdgcn_raw_buffer_load_b32(rs, off[4 + a], so, 0);
        }
    };
    auto transform = [&](const Patch &r, float (&v)[16]) {
        float d[16];
        if (ABL & 4) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                v[4 * a] = __builtin_bit_cast(float, r.p0[a]); v[4 * a + 1] = __builtin_bit_cast(float, r.p1[a]);
                v[4 * a + 2] = __builtin_bit_cast(float, r.h[a]); v[4 * a + 3] = __builtin_bit_cast(float, r.p0[a] ^ r.h[a]);
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int p0 = (int)r.p0[a], hv = (int)r.h[a];
            const int p1 = (LOAD == 2 && !p1ok) ? 0 : (int)r.p1[a];
            d[4 * a] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hv, p1, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
            d[4 * a + 1] = __builtin_bit_cast(float, p0);
            d[4 * a + 2] = __builtin_bit_cast(float, p1);
            d[4 * a + 3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hv, p0, 0x101 /* row_shl:1 */, 0xf, 0xf, false));
        }
        float t[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = d[j] - d[8 + j];
            t[4 + j] = d[4 + j] + d[8 + j];
            t[8 + j] = d[8 + j] - d[4 + j];
            t[12 + j] = d[12 + j] - d[4 + j];               // = -(B^T d)[3][j]: the packed filters carry the same sign
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[4 * i] = t[4 * i] - t[4 * i + 2];
            v[4 * i + 1] = t[4 * i + 1] + t[4 * i + 2];
            v[4 * i + 2] = t[4 * i + 2] - t[4 * i + 1];
            v[4 * i + 3] = t[4 * i + 1] - t[4 * i + 3];
        }
    };
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto uload = [&](f4 &u0, f4 &u1, int c) {
        const unsigned so = (unsigned)c * ustep;
        u0 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, uoff, so, 0));
        u1 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, uoff + 4096u, so, 0));
    };
    auto ustore = [&](int st, const f4 &u0, const f4 &u1) {
        f4 *d0 = (f4 *)Us[st] + tid;
        d0[0] = u0; d0[256] = u1;
    };

    f32x4 acc[16][NBLK];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < NBLK; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one K step: MFMAs of chunk c (operands vc, slab Us[c & 1]) with, underneath them, the transform of chunk c + 1
    // (patch pin -> vn), the patch prefetch of chunk c + 2 (-> pout), the LDS store of slab c + 1 and the slab
    // prefetch of chunk c + 2
    auto kstep = [&](int c, const Patch &pin, Patch &pout, const float (&vc)[16], float (&vn)[16], f4 &u0, f4 &u1) {
        f4 a[8];
        const f4 *us = (const f4 *)Us[c & 1] + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = us[i * 64];
        transform(pin, vn);
        const int cn = min(c + 2, nchunks - 1);      // prefetch index clamped (scalar): never reads past the tensors
        gload(pout, cn);
        ustore((c + 1) & 1, u0, u1);
        uload(u0, u1, cn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int k = 0; k < NBLK; ++k) acc[4 * q + e][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * NBLK + k][e], vc[4 * q + e], acc[4 * q + e][k], 0, 0, 0);
            }
        }
        // pinned interleave: the 8 operand reads first, a dozen transform instructions while they are in flight, then one
        // MFMA : one or two other instructions
        if (MODE == 0) {
        WN_SGB(WN_M_DS_RD, 8);
        WN_SGB(WN_M_VALU, 12);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            WN_SGB(WN_M_MFMA, 1);
            if (i < 16) WN_SGB(WN_M_VALU, 2); else WN_SGB(WN_M_VALU, 1);
            if (i % 3 == 1 && i < 30) WN_SGB(WN_M_VMEM_RD, 1);
            if (i == 20 || i == 24) WN_SGB(WN_M_DS_WR, 1);
        }
        }
        if (!(ABL & 2)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);          // the next step's transform must not be hoisted above its patch loads' latency
    };

    Patch pa, pb;
    float va[16], vb[16];
    f4 u0, u1;
    gload(pa, 0);
    uload(u0, u1, 0);
    gload(pb, min(1, nchunks - 1));
    ustore(0, u0, u1);
    uload(u0, u1, min(1, nchunks - 1));
    transform(pa, va);
    __syncthreads();
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        kstep(c, pb, pa, va, vb, u0, u1);          // chunk c: transforms c+1 (in pb) -> vb, prefetches c+2 -> pa
        kstep(c + 1, pa, pb, vb, va, u0, u1);      // chunk c+1: transforms c+2 (in pa) -> va, prefetches c+3 -> pb
    }
    if (nchunks & 1) kstep(c, pb, pa, va, vb, u0, u1);   // odd chunk count: one more step (its look-ahead work is discarded)
    if (ABL & 1) {
        f32x4 t = acc[0][0];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int k = 0; k < NBLK; ++k) if (i | k) t += acc[i][k];
        if (t[0] + t[1] + t[2] + t[3] == 12345.678f) y[tid] = t[0];
        return;
    }
    wn_epilogue<POOL, LOAD, NBLK>(acc, bias, y, residual, Cout, H, W, act, b, cb, kq, ty, tx);
}

// ---------------------------------------------------------------------------------------------------------------------
// Shared-transform variant.  Measured on gfx950 (tools/ubench/mfma_valu.hip): next to v_mfma_f32_16x16x4_f32 a VALU
// instruction is NOT hidden -- it costs ~3 cycles with two wavefronts per SIMD (5.5 with one), s_nop / SALU cost nothing:
// the f32 MFMA and the vector ALU serialise, so   time = 32 * n_mfma + 3 * n_valu   and the only lever left is the VALU
// count.  Ablations of the pipelined kernel on conv1b: input transform 20 %, patch loads 14 %, output transform 11 %.
//   * a workgroup = 2 tile rows x 2 cout slices of 32: the two wavefronts that own the same 16 tiles each transform HALF of
//     every 4x4 patch (two of the four Winograd rows: 16 adds instead of 32, 3 patch rows instead of 4) and exchange the
//     halves through LDS in MFMA operand order (2 ds_write_b128 + 4 ds_read_b128 per chunk, which the MFMAs do hide);
//   * slice 0 computes rows {0, 1} from patch rows (0, 1, 2), slice 1 rows {3, 2} from patch rows (3, 2, 1) with the SAME
//     instruction stream:  first = dA - dC,  second = dB + sgn * dC  (sgn = +1 / -1, one fma) -- slice 1's first row is
//     -(B^T d)[3], which is why row 3 is carried negated in the packed filters as well (products unchanged);
//   * the bias is folded into the accumulator of Winograd position (1, 1), whose output-transform weight is +1 for all four
//     outputs: no bias instruction in the epilogue.
// K loop software-pipelined and pinned like the pipelined kernel; one barrier per chunk.
// ---------------------------------------------------------------------------------------------------------------------
template <bool POOL, int LOAD, int ABL = 0, int ACT = -1>
__global__ void __launch_bounds__(256, 2) wino_conv3x3_shared_kernel(
    const float *__restrict__ x, const float *__restrict__ upk, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncb, int ncb32, int act)
{
    constexpr int NBLK = 2;
    constexpr int SLAB = 1024 * NBLK;
    static_assert(LOAD == 1 || LOAD == 2, "paired loads only");
    __shared__ __attribute__((aligned(16))) float Us[2][2][SLAB];          // [buffer][cout slice]           32 KB
    __shared__ __attribute__((aligned(16))) float Vs[2][2][1024];          // [buffer][tile row][q][lane][e] 16 KB
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int cb = jj % ncb;                                                // 64-cout block
    const int sl = jj / ncb;
    const int s = xcd * Sx + sl;
    if (sl >= Sx || s >= S) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = w >> 1, cs = w & 1;
    const int col = lane & 15, kq = lane >> 4;
    int b, ty, tx;
    if (LOAD == 2) {
        // odd W: LINEAR tiling.  The tile grid of an image (nbx = tiles per row) is one sequence, a wavefront takes 16 consecutive
        // tiles wherever the rows break (nby = 32-tile groups per image) -- a 67-pixel row has 34 tiles, and 16-wide blocks would
        // compute 48.  The neighbour exchange by DPP stays correct across a row break without any extra instruction BECAUSE W is
        // odd: the last tile of a row has its second column outside the image (forced to 0 below), which is exactly the zero
        // padding the first tile of the next row needs as left neighbour; and what the last tile receives from the next row's
        // first tile only reaches its second output column (x = W), which does not exist and is never stored.
        const int g = s % nby;
        b = s / nby;
        const int t = g * 32 + tr * 16 + col;
        ty = t / nbx; tx = t - ty * nbx;
        if (ty >= ((H + 1) >> 1)) { ty = H; tx = 0; }                      // past the last tile: every load out of range, nothing stored
    } else {
        const int bx = s % nbx, by = (s / nbx) % nby;
        b = s / (nbx * nby);
        ty = by * 2 + tr; tx = bx * WN_TX + col;
    }
    const int HW = H * W;
    const int nchunks = Cin >> 2;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)b * Cin * HW), 0, Cin * HW * 4, WN_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void *)upk, 0, nchunks * ncb32 * SLAB * 4, WN_RSRC_FLAGS);
    // patch rows this wavefront needs: slice 0 rows (0, 1, 2), slice 1 rows (3, 2, 1) of the 4x4 patch
    unsigned off[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int pr = cs ? 3 - a : a;
        const int iy = 2 * ty - 1 + pr;
        const bool rv = iy >= 0 && iy < H;
        const unsigned rowb = (unsigned)(kq * HW + iy * W) * 4u;
        off[a] = (rv && 2 * tx < W) ? rowb + 8u * tx : WN_OOB;
        const int hx = (col == 0) ? 2 * tx - 1 : 2 * tx + 2;
        off[3 + a] = (rv && (col == 0 || col == 15) && hx >= 0 && hx < W) ? rowb + 4u * hx : WN_OOB;
    }
    const float sgn = cs ? -1.f : 1.f;
    const bool p1ok = 2 * tx + 1 < W;
    const unsigned cstep = 16u * HW;
    // packed filters: the workgroup stages the two 32-cout slabs of its 64-cout block; thread tid moves float4 tid and tid + 256 of each
    const unsigned uoff = (unsigned)(2 * cb * SLAB + tid * 4) * 4u;
    const unsigned ustep = (unsigned)ncb32 * SLAB * 4u;
    const bool s1ok = 2 * cb + 1 < ncb32;                                   // odd number of 32-cout blocks: the last slice 1 does not exist

    struct Patch { unsigned p0[3], p1[3], h[3]; };
    auto gload = [&](Patch &r, int c) {
        const unsigned so = (unsigned)c * cstep;
        if (ABL & 8) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { r.p0[a] = so + a; r.p1[a] = so ^ a; r.h[a] = so; }
            return;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const auto pr = __builtin_amdgcn_raw_buffer_load_b64(rs, off[a], so, 0);
            r.p0[a] = pr[0]; r.p1[a] = pr[1];
            r.h[a] = __builtin_amdgcn_raw_buffer_load_b32(rs, off[3 + a], so, 0);
        }
    };
    // half transform: o[0..3] = "first" Winograd row, o[4..7] = "second" row of this slice
    auto htransform = [&](const Patch &r, float (&o)[8]) {
        float d[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int p0 = (int)r.p0[a], hv = (int)r.h[a];
            const int p1 = (LOAD == 2 && !p1ok) ? 0 : (int)r.p1[a];
            d[a][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hv, p1, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
            d[a][1] = __builtin_bit_cast(float, p0);
            d[a][2] = __builtin_bit_cast(float, p1);
            d[a][3] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hv, p0, 0x101 /* row_shl:1 */, 0xf, 0xf, false));
        }
        float tf[4], ts[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tf[j] = d[0][j] - d[2][j];
            ts[j] = __builtin_fmaf(d[2][j], sgn, d[1][j]);
        }
        o[0] = tf[0] - tf[2]; o[1] = tf[1] + tf[2]; o[2] = tf[2] - tf[1]; o[3] = tf[1] - tf[3];
        o[4] = ts[0] - ts[2]; o[5] = ts[1] + ts[2]; o[6] = ts[2] - ts[1]; o[7] = ts[1] - ts[3];
    };
    typedef float f4 __attribute__((ext_vector_type(4)));
    // slice 0 owns q = 0 (first), 1 (second); slice 1 owns q = 3 (first), 2 (second)
    const int qf = cs ? 3 : 0, qs = cs ? 2 : 1;
    auto vstore = [&](int st, const float (&o)[8]) {
        f4 *dv = (f4 *)Vs[st][tr] + lane;
        dv[qf * 64] = (f4){o[0], o[1], o[2], o[3]};
        dv[qs * 64] = (f4){o[4], o[5], o[6], o[7]};
    };
    struct UReg { f4 u[4]; };
    auto uload = [&](UReg &r, int c) {
        const unsigned so = (unsigned)c * ustep;
        r.u[0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, uoff, so, 0));
        r.u[1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, uoff + 4096u, so, 0));
        r.u[2] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, s1ok ? uoff + 8192u : WN_OOB, so, 0));
        r.u[3] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(ru, s1ok ? uoff + 12288u : WN_OOB, so, 0));
    };
    auto ustore = [&](int st, const UReg &r) {
        f4 *d0 = (f4 *)Us[st][0] + tid;                                     // the two slabs are contiguous: [0] then [1]
        d0[0] = r.u[0]; d0[256] = r.u[1]; d0[512] = r.u[2]; d0[768] = r.u[3];
    };

    f32x4 acc[16][NBLK];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < NBLK; ++k) acc[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias) {                                                             // position (1, 1) reaches all four outputs with weight +1
#pragma unroll
        for (int k = 0; k < NBLK; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cb * 64 + cs * 32 + k * 16 + 4 * kq + r;
                acc[5][k][r] = co < Cout ? bias[co] : 0.f;
            }
    }

    auto kstep = [&](int c, const Patch &pin, Patch &pout, UReg &ur) {
        f4 a[8], vv[4];
        const f4 *us = (const f4 *)Us[c & 1][cs] + lane;
        const f4 *vs = (const f4 *)Vs[c & 1][tr] + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) vv[q] = vs[q * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = us[i * 64];
        float o[8];
        if (!(ABL & 4)) htransform(pin, o);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = __builtin_bit_cast(float, pin.p0[i % 3] ^ pin.h[i % 3] + i);
        }
        vstore((c + 1) & 1, o);
        const int cn = min(c + 2, nchunks - 1);      // prefetch index clamped (scalar): never reads past the tensors
        gload(pout, cn);
        if (!(ABL & 16)) { ustore((c + 1) & 1, ur); uload(ur, cn); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int k = 0; k < NBLK; ++k) acc[4 * q + e][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q * NBLK + k][e], vv[q][e], acc[4 * q + e][k], 0, 0, 0);
            }
        }
        // pinned interleave: 12 operand reads, the first transform instructions while they fly, then MFMA : 1 other
        WN_SGB(WN_M_DS_RD, 12);
        WN_SGB(WN_M_VALU, 6);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            WN_SGB(WN_M_MFMA, 1);
            WN_SGB(WN_M_VALU, 1);
            if (i % 3 == 0 && i < 30) WN_SGB(WN_M_VMEM_RD, 1);
            if (i == 12 || i == 14) WN_SGB(WN_M_DS_WR, 1);                  // the two halves of V (needed first by the partner)
            if (i >= 20 && i < 28 && (i & 1) == 0) WN_SGB(WN_M_DS_WR, 1);   // the next chunk's filter slabs
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };

    Patch pa, pb;
    UReg ur;
    gload(pa, 0);
    uload(ur, 0);
    gload(pb, min(1, nchunks - 1));
    {
        float o[8];
        htransform(pa, o);
        vstore(0, o);
    }
    ustore(0, ur);
    uload(ur, min(1, nchunks - 1));
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // (a variant that prefetched the patches two chunks ahead -- three rotating register sets, loop unrolled by six -- hit the
    // 256-register limit and measured slower: 8.78 vs 8.27 ms on conv1b)
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
        kstep(c, pb, pa, ur);
        kstep(c + 1, pa, pb, ur);
    }
    if (nchunks & 1) kstep(c, pb, pa, ur);              // odd chunk count: one more step (its look-ahead work is discarded)
    if (ABL & 1) {
        f32x4 t = acc[0][0];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int k = 0; k < NBLK; ++k) if (i | k) t += acc[i][k];
        if (t[0] + t[1] + t[2] + t[3] == 12345.678f) y[tid] = t[0];
        return;
    }
    wn_epilogue<POOL, LOAD, NBLK, false, ACT>(acc, nullptr, y, residual, Cout, H, W, act, b, 2 * cb + cs, kq, ty, tx);
}

static inline int wn_coutp(int Cout, int nblk) { const int q = 16 * nblk; return (Cout + q - 1) / q * q; }

extern "C" {

size_t mfr_wino_filter_bytes(int Cin, int Cout)
{
    if (Cin <= 0 || Cout <= 0 || (Cin & 3)) return 0;
    return sizeof(float) * 16 * (size_t)Cin * wn_coutp(Cout, wn_nblk(Cout));
}

int mfr_wino_filter_transform(const float *w, int Cin, int Cout, float *upk, void *stream)
{
    if (!w || !upk || Cin <= 0 || Cout <= 0 || (Cin & 3)) return MFR_E_ARG;
    const int nblk = wn_nblk(Cout), CoutP = wn_coutp(Cout, nblk);
    if (CoutP != Cout && mfr_zero_async(upk, sizeof(float) * 16 * (size_t)Cin * CoutP, (hipStream_t)stream) != hipSuccess) return MFR_E_LAUNCH;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, CoutP, nblk, upk);
    CHECK_LAUNCH();
    return 0;
}

// variant: 0 = default (the faster measured one for the shape), 1 = classic (transform then MFMA run per chunk),
// 2 = software-pipelined (transform of chunk c+1 under the MFMAs of chunk c), 3 = 2 without the pinned interleave,
// 4 = shared-transform kernel (two cout slices split every patch transform; see wino_conv3x3_shared_kernel).  All variants give
// bit-identical results; the selector exists for the A/B timing tool and the parity tests.
int mfr_conv3x3_wino_variant(const float *x, const float *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                             int act, int pool, int variant, float *y, void *stream)
{
    if (!x || !upk || !y || B <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 3) || H <= 0 || W <= 0 || act < 0 || act > 2) return MFR_E_ARG;
    if (pool && (H < 2 || W < 2 || residual)) return MFR_E_ARG;
    if (variant < 0 || (variant > 4 && (variant < 10 || variant > 33)) ) return MFR_E_ARG;
    if ((size_t)4 * Cin * H * W >= 0x7fffffffull) return MFR_E_ARG;    // one image must fit a 2 GB buffer descriptor
    const int nblk = WN_NBLK;
    int nbx = ((W + 1) / 2 + WN_TX - 1) / WN_TX;
    const int nby = ((H + 1) / 2 + WN_TY - 1) / WN_TY;
    const int ncb = wn_coutp(Cout, nblk) / (16 * nblk);
    if ((size_t)16 * Cin * wn_coutp(Cout, nblk) * 4 >= 0x7fffffffull) return MFR_E_ARG;
    const long long S = (long long)nbx * nby * B;
    const long long Sx = (S + 7) / 8;                        // spatial blocks per XCD
    const long long grid = Sx * 8 * ncb;
    if (grid > 0x7fffffffll) return MFR_E_ARG;
    // default: the shared-transform kernel when the cout count fills whole 64-channel workgroups (an odd number of 32-channel
    // blocks would idle one slice of the last block), else the pipelined kernel
    if (variant == 0) variant = ((wn_coutp(Cout, nblk) / 32) & 1) ? 2 : 4;
    const bool odd = W & 1;
    hipStream_t st = (hipStream_t)stream;
#define WN_ARGS x, upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncb, act
#define WN_GO(K) hipLaunchKernelGGL(K, dim3((unsigned)grid), dim3(256), 0, st, WN_ARGS)
    if (variant == 4 || variant >= 26) {                   // 27..33: timing ablations of the shared kernel                   // shared-transform kernel: workgroup = 2 tile rows x 64 couts
        const int ncb32 = wn_coutp(Cout, nblk) / 32, ncb64 = (ncb32 + 1) / 2;
        // even W: blocks of 2 tile rows x 16 tile columns; odd W: linear tiling, nbx = tiles per row, nby2 = 32-tile groups per image
        const int txn = (W + 1) / 2, tyn = (H + 1) / 2;
        const int nby2 = odd ? (int)(((long long)txn * tyn + 31) / 32) : (tyn + 1) / 2;
        if (odd) nbx = txn;
        const long long S2 = odd ? (long long)nby2 * B : (long long)nbx * nby2 * B, Sx2 = (S2 + 7) / 8, grid2 = Sx2 * 8 * ncb64;
        if (grid2 > 0x7fffffffll) return MFR_E_ARG;
#define WN_GO2(K) hipLaunchKernelGGL(K, dim3((unsigned)grid2), dim3(256), 0, st, x, upk, bias, y, residual, Cin, Cout, H, W, nbx, nby2, (int)S2, (int)Sx2, ncb64, ncb32, act)
        if (variant == 4) {
#define WN_GO2_ACT(P, L) do { if (act == 1) WN_GO2((wino_conv3x3_shared_kernel<P, L, 0, 1>)); else if (act == 2) WN_GO2((wino_conv3x3_shared_kernel<P, L, 0, 2>)); \
                              else WN_GO2((wino_conv3x3_shared_kernel<P, L, 0, 0>)); } while (0)
            if (pool) { if (odd) WN_GO2_ACT(true, 2); else WN_GO2_ACT(true, 1); }
            else { if (odd) WN_GO2_ACT(false, 2); else WN_GO2_ACT(false, 1); }
#undef WN_GO2_ACT
        } else if (variant == 27) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 1>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 1>)); }
        else if (variant == 28) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 4>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 4>)); }
        else if (variant == 29) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 5>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 5>)); }
        else if (variant == 30) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 8>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 8>)); }
        else if (variant == 31) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 16>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 16>)); }
        else if (variant == 32) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 29>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 29>)); }
        else if (variant == 33) { if (pool) WN_GO2((wino_conv3x3_shared_kernel<true, 1, 13>)); else WN_GO2((wino_conv3x3_shared_kernel<false, 1, 13>)); }
        else return MFR_E_ARG;
#undef WN_GO2
    } else if (variant == 2) {
        if (pool) { if (odd) WN_GO((wino_conv3x3_pipe_kernel<true, 2, 0>)); else WN_GO((wino_conv3x3_pipe_kernel<true, 1, 0>)); }
        else { if (odd) WN_GO((wino_conv3x3_pipe_kernel<false, 2, 0>)); else WN_GO((wino_conv3x3_pipe_kernel<false, 1, 0>)); }
    } else if (variant >= 10) {                            // timing ablations of the pipelined kernel (even W, as launched by the tool)
        switch (variant - 10) {
#define WN_ABL(A) case A: if (pool) WN_GO((wino_conv3x3_pipe_kernel<true, 1, 0, A>)); else WN_GO((wino_conv3x3_pipe_kernel<false, 1, 0, A>)); break;
        WN_ABL(1) WN_ABL(2) WN_ABL(3) WN_ABL(4) WN_ABL(8) WN_ABL(12) WN_ABL(15)
#undef WN_ABL
        default: return MFR_E_ARG;
        }
    } else if (variant == 3) {
        if (pool) { if (odd) WN_GO((wino_conv3x3_pipe_kernel<true, 2, 1>)); else WN_GO((wino_conv3x3_pipe_kernel<true, 1, 1>)); }
        else { if (odd) WN_GO((wino_conv3x3_pipe_kernel<false, 2, 1>)); else WN_GO((wino_conv3x3_pipe_kernel<false, 1, 1>)); }
    } else {
        if (pool) { if (odd) WN_GO((wino_conv3x3_kernel<true, 2, WN_NBLK>)); else WN_GO((wino_conv3x3_kernel<true, 1, WN_NBLK>)); }
        else { if (odd) WN_GO((wino_conv3x3_kernel<false, 2, WN_NBLK>)); else WN_GO((wino_conv3x3_kernel<false, 1, WN_NBLK>)); }
    }
#undef WN_GO
#undef WN_ARGS
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv3x3_wino(const float *x, const float *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                     int act, int pool, float *y, void *stream)
{
    return mfr_conv3x3_wino_variant(x, upk, bias, residual, B, Cin, Cout, H, W, act, pool, 0, y, stream);
}

}  // extern "C"
