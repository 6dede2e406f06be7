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
#include <stdlib.h>

#include "../../include/mfr_hip.h"

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

// NBLK (16-cout MFMA blocks per workgroup).  Measured on the SuperPoint layers (B = 32 images): NBLK = 2 (two
// workgroups per CU) 12.2 ms for the nine layers, NBLK = 4 (one wavefront per SIMD) 14.3 ms -- the un-overlapped
// prologue/epilogue and barrier skew of a lone wavefront cost more than the halved transform work saves, so 2 is
// the default; MFR_WINO_NBLK=4 selects the other instantiation (tuning aid, needs Cout % 64 == 0).
static inline int wn_nblk(int Cout)
{
    const char *ev = getenv("MFR_WINO_NBLK");
    return (ev && atoi(ev) == 4 && Cout % 64 == 0) ? 4 : 2;
}

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
        // xi = 4 r + j  ->  q = r, e = j
        *(float4 *)(dst + (((size_t)r * nblk + blk) * 64 + kq * 16 + row) * 4) = make_float4(u[0], u[1], u[2], u[3]);
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
    constexpr int WCO = 16 * NBLK;                           // output channels per workgroup
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
            t[12 + j] = d[4 + j] - d[12 + j];
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
            const float bv = bias ? bias[co] : 0.f;
            float y00 = a0[0] + a0[1] + a0[2] + bv, y01 = a0[1] - a0[2] - a0[3] + bv;
            float y10 = a1[0] + a1[1] + a1[2] + bv, y11 = a1[1] - a1[2] - a1[3] + bv;
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

template <bool POOL, int LOAD, int NBLK>
static void wn_launch(long long grid, hipStream_t st, const float *x, const float *upk, const float *bias, float *y, const float *residual,
                      int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncb, int act)
{
    hipLaunchKernelGGL((wino_conv3x3_kernel<POOL, LOAD, NBLK>), dim3((unsigned)grid), dim3(256), 0, st, x, upk, bias, y, residual, Cin, Cout,
                       H, W, nbx, nby, S, Sx, ncb, act);
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
    if (CoutP != Cout && hipMemsetAsync(upk, 0, sizeof(float) * 16 * (size_t)Cin * CoutP, (hipStream_t)stream) != hipSuccess) return MFR_E_LAUNCH;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, CoutP, nblk, upk);
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv3x3_wino(const float *x, const float *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                     int act, int pool, float *y, void *stream)
{
    if (!x || !upk || !y || B <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 3) || H <= 0 || W <= 0 || act < 0 || act > 2) return MFR_E_ARG;
    if (pool && (H < 2 || W < 2 || residual)) return MFR_E_ARG;
    if ((size_t)4 * Cin * H * W >= 0x7fffffffull) return MFR_E_ARG;    // one image must fit a 2 GB buffer descriptor
    const int nblk = wn_nblk(Cout);
    const int nbx = ((W + 1) / 2 + WN_TX - 1) / WN_TX, nby = ((H + 1) / 2 + WN_TY - 1) / WN_TY;
    const int ncb = wn_coutp(Cout, nblk) / (16 * nblk);
    const long long S = (long long)nbx * nby * B;
    const long long Sx = (S + 7) / 8;                        // spatial blocks per XCD
    const long long grid = Sx * 8 * ncb;
    if (grid > 0x7fffffffll) return MFR_E_ARG;
    const char *ev = getenv("MFR_WINO_LOAD");                // tuning aid: 0 forces the 16-dword path
    const int load = (ev && atoi(ev) == 0) ? 0 : ((W & 1) ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
#define WN_ARGS grid, st, x, upk, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncb, act
#define WN_PICK_LOAD(P, N) do { if (load == 0) wn_launch<P, 0, N>(WN_ARGS); else if (load == 1) wn_launch<P, 1, N>(WN_ARGS); \
                                else wn_launch<P, 2, N>(WN_ARGS); } while (0)
    if (nblk == 4) { if (pool) WN_PICK_LOAD(true, 4); else WN_PICK_LOAD(false, 4); }
    else { if (pool) WN_PICK_LOAD(true, 2); else WN_PICK_LOAD(false, 2); }
#undef WN_PICK_LOAD
#undef WN_ARGS
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
