// winograd_split.hip -- 3x3 / stride 1 / pad 1 convolution (NCHW f32 in, f32 out) as a fused Winograd F(2x2, 3x3) kernel on the gfx950
// 16-bit matrix cores at fp32 accuracy (operand splitting), with the layer epilogue (+bias, activation, optional 2x2 max-pool, optional
// residual) folded into the output transform.
//
// Reference call site: SuperGlue_matcher / LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-120) -> the un-vendored
// SuperPoint encoder (conv1b..conv4b, convPa, convDa) and LoFTR ResNet-FPN backbone.
//
// Arithmetic.  Y = A^T [ (G g G^T) (.) (B^T d B) ] A per (cin, cout): 16 independent GEMMs over cin, one per Winograd position
// (i, j).  The transformed filter U = G g G^T (fp64, rounded once to fp32) and the transformed patch V = B^T d B (fp32, exactly as the
// exact-fp32 kernel winograd_conv.hip forms it) are the operands of those GEMMs; two arithmetics (template parameter F16):
//   f16x2  (round 5, the default; split_f16.h)  V as two f16 terms (the low one scaled by 2^11), U pre-scaled per OUTPUT channel and packed as
//          three f16 terms (uh, ul, uh 2^-11): THREE v_mfma_f32_32x32x16_f16 per product block, 2.5 VALU per V element; the output transform
//          multiplies by the channel's 1 / scale.  Precondition |V| <= 65504, i.e. |activation| < 16376.
//   bf16x3 (rounds 3-4)  both operands split EXACTLY into three bf16 terms, six partial products, 5.5 VALU per V element.
// Either way the error against an fp64 product is that of the exact-fp32 matrix instruction (profiles/r05_f16x2_probe.jsonl,
// r03_bf16x3_probe.jsonl; the layer tests hold both to the same 2e-5 bar against a float64 convolution).
//
// Mapping to CDNA4 (round 4's eight-wavefront kernel; its two predecessors -- one wavefront per SIMD, two independent 4-wavefront
// workgroups per CU -- measured no faster and left the library in round 5: profiles/r04_ab_conv.json, r04_ablate_conv_*.json):
//   * a workgroup = 64 tiles (16 x 4) x 64 output channels, EIGHT wavefronts (two per SIMD); wavefront w owns Winograd row i = w >> 1 and
//     the column pair j in {2 jp, 2 jp + 1}, jp = w & 1: 2 positions x 2 channel blocks x 2 tile blocks = 128 accumulators.  Every
//     position's V is produced by exactly one wavefront and IS the B operand of its MFMAs; every filter fragment is requested by exactly
//     one wavefront (12 per K step, resident, re-requested after their last use) and streams from L2 straight into registers
//     (pre-split, pre-packed in operand order).  No operand crosses LDS.
//   * the raw input patches (10 rows x 34 columns x 16 channels per K step) are the only thing staged in LDS: LDS-DMA
//     (`buffer_load_dwordx4 ... lds`, issued from inline asm -- hipcc treats a DMA it knows about as a pending store to the whole LDS array
//     and waits vmcnt(0) before the next ds_read) a whole K step ahead, no register; everything outside the image arrives as zeros
//     (offset beyond the buffer), the few pieces that STRADDLE the left / right image border are patched in LDS by the lane that
//     requested them.  The staged row stride (48 floats) puts the two tile rows a wavefront reads together on disjoint banks.
//   * the column pair decides which three of the four patch columns a wavefront needs -- jp = 0: 0, 1, 2; jp = 1: 1, 2, 3 -- fetched as one
//     8-byte and one 4-byte LDS read whose OFFSETS depend on jp, so that the code is the same for both: X, Y = columns (0, 1) | (2, 3),
//     Z = column 2 | 1, V(jj = 0) = X - Z, V(jj = 1) = Z + beta Y with beta = +1 | -1.
//   * the output transform needs all four rows and both column pairs of a tile: two rounds (one per 32-channel block) through the
//     128 KB the patch stages no longer need; wavefront q finishes tile block q & 1, channel group q >> 1 of the round.
//   * grid: 1-D, XCD-aware (every XCD walks one contiguous raster range of spatial blocks in chunks, see the kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"
#include "split_f16.h"
#include "guard.h"

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
#define WB_FRAGS_PER_KSTEP 96      // 4 i x 4 j x 2 cout blocks x 3 terms (both arithmetics)

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
// filters: w [Cout, Cin, 3, 3] f32 -> U = G g G^T (fp64 arithmetic, rounded once to fp32), split into three 16-bit terms and
// packed as MFMA A operands: fragment f = ((((cg * nks + c) * 4 + i) * 4 + j) * 2 + mb) * 3 + term, 64 lanes x 16 bytes;
// lane l holds cout cg*64 + mb*32 + (l & 31), input channels 16 c + 8 (l >> 5) + (0..7).  Channels beyond Cin / Cout are zero.
// f16x2: the blob ends with the per-output-channel 1 / scale (ncg x 64 floats, written by wb_filter_scale_kernel BEFORE the pack kernel runs).
__device__ __forceinline__ float wb_u_value(const float *__restrict__ g, int i, int j)
{
    const double G[4][3] = { { 1.0, 0.0, 0.0 }, { 0.5, 0.5, 0.5 }, { 0.5, -0.5, 0.5 }, { 0.0, 0.0, 1.0 } };
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) s += G[i][a] * (double)g[3 * a + b] * G[j][b];
    return (float)s;
}
__global__ void __launch_bounds__(64) wb_filter_scale_kernel(const float *__restrict__ w, int Cin, int Cout, int cpad, float *__restrict__ oscale)
{
    const int co = blockIdx.x, lane = threadIdx.x;
    float mx = 0.f;
    if (co < Cout)
        for (int t = lane; t < Cin * 16; t += 64) {
            const int ci = t >> 4, p = t & 15;
            mx = fmaxf(mx, fabsf(wb_u_value(w + ((size_t)co * Cin + ci) * 9, p >> 2, p & 3)));
        }
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0 && co < cpad) oscale[co] = 1.0f / sf_feature_scale(mx);
}
template <bool F16>
__global__ void __launch_bounds__(256) wb_filter_kernel(const float *__restrict__ w, int Cin, int Cout, int nks, long long total, const float *__restrict__ oscale, uint4 *__restrict__ upk)
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
    const float s = F16 ? 1.0f / oscale[co] : 1.0f;
    unsigned word[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = 16 * c + 8 * (l >> 5) + e;
        float u = 0.f;
        if (co < Cout && ci < Cin) u = wb_u_value(w + ((size_t)co * Cin + ci) * 9, i, j);
        if (F16) {
            unsigned short uh, ul, uq;
            sf_split_w(u * s, uh, ul, uq);
            word[e] = (unsigned)(term == 0 ? uh : term == 1 ? ul : uq) << 16;
        } else {
            unsigned h, m, lo;
            wb_split3(u, h, m, lo);
            word[e] = term == 0 ? h : term == 1 ? m : lo;
        }
    }
    upk[t] = make_uint4(wb_pack(word[0], word[1]), wb_pack(word[2], word[3]), wb_pack(word[4], word[5]), wb_pack(word[6], word[7]));
}

// ---------------------------------------------------------------------------------------------------------------------
#define WB_MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// Measurement build only (tools/conv_timeline.py compiles THIS file a second time with -DWB_PROF into tools/ubench/libwino_prof.so; the product
// library never defines it): s_memtime stamps of the eight wavefronts of one workgroup in the middle of the grid.
#ifdef WB_PROF
__device__ unsigned long long wb_prof[8][32];
#define WB_STAMP(k) do { if (blockIdx.x == (gridDim.x / 2 | 5u) && lane == 0) { __builtin_amdgcn_sched_barrier(0); wb_prof[w][(k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define WB_STAMP(k) do { } while (0)
#endif

// What bounds this kernel (profiles/r05_pmc_conv1b.json, r05_conv1b_timeline.json): not memory (traffic 1.01 x algorithmic) and not the matrix pipe
// (20 % busy with f16x2) but the SIMD's issue port -- a VALU / LDS / VMEM instruction costs ~4 cycles of the SIMD whichever wavefront issues it
// and does NOT overlap an MFMA's 32: a K step costs a wavefront 24 x 32 + ~300 x 4 cycles, two wavefronts per SIMD, 3.7 - 4.6 k cycles measured
// (219 VALU + 72 LDS + 16 VMEM instructions per 24 MFMAs in the loop body).  Around the loop: 17 % of a conv1b workgroup's time is the wait for
// the first patches (HBM latency, nothing else resident on the CU) and 23 % the output transform, whose exchange is bound by the LDS store path
// (128 KB per round through ds_write_b128).
// Tried in round 5 and dropped (profiles/r05_ab_conv_128ch_workgroup.json): a 32-tile x 128-channel workgroup for layers with more than 64 output
// channels (one tile block per wavefront instead of two: half the patch reads, row combinations and V splits per MFMA, uq formed in registers so
// that four channel blocks of fragments fit).  Parity-green on every shape, but 6 - 11 % SLOWER on every layer (LoFTR 128 -> 128 at 360x272:
// 2.55 -> 2.83 ms, 196 -> 196: 6.94 -> 7.38): twice the filter fragments per MFMA from L2, 2.25 x instead of 1.9 x patch read amplification and
// four output rounds instead of two cost more than the saved VALU.
template <bool POOL, bool F16>
__global__ void __launch_bounds__(512, 2) wino_split_p8_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ upk, const float *__restrict__ oscale, const float *__restrict__ bias, float *__restrict__ y,
    const float *__restrict__ residual, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int Sx, int ncg, int nks, int act, int chunk, int *guard)
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

    // ---- patch staging by LDS-DMA: per K step and channel a 10-row x 48-column window (columns 32 bx - 1 + cx, rows 8 by - 1 + r; 34 x 10
    // are used), fetched as 16-byte pieces: piece = lane (60 of 64 lanes), 5 rows x 12 pieces per instruction, two instructions per
    // channel; this wavefront stages input channels 2 w, 2 w + 1 of the step's 16
    const int prow = lane / 12, pk = lane - 12 * prow;      // lanes 60..63: prow = 5 -> no piece
    const int pix = 32 * bx - 1 + 4 * pk;                   // first column of this lane's piece
    unsigned voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int iy = 8 * by - 1 + 5 * h + prow;
        const bool ok = prow < 5 && pk < 9 && iy >= 0 && iy < H && pix < W && pix + 3 >= 0;
        voff[h] = ok ? (unsigned)(iy * W + max(pix, 0)) * 4u : WB_OOB;   // the piece that starts at column -1 is fetched from column 0 and shifted right by one in LDS (pfix)
    }
    const bool fixl = pix < 0;
    unsigned keep = 0;                                      // which of the 4 dwords lie inside the row
#pragma unroll
    for (int d = 0; d < 4; ++d) if (pix + d >= 0 && pix + d < W) keep |= 1u << d;
    const bool edge = (bx == 0) || (32 * bx + 35 >= W);     // wave-uniform: some piece of this workgroup straddles a border
    typedef unsigned wb_u32x4 __attribute__((ext_vector_type(4)));
    wb_u32x4 xdesc;
    {
        const unsigned long long xa = (unsigned long long)(x + (size_t)b * Cin * HW);
        xdesc.x = __builtin_amdgcn_readfirstlane((unsigned)xa);
        xdesc.y = __builtin_amdgcn_readfirstlane((unsigned)(xa >> 32) & 0xffffu);
        xdesc.z = (unsigned)(Cin * HW) * 4u;                // channels >= Cin lie beyond the buffer: zeros
        xdesc.w = WB_RSRC_FLAGS;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float *)lds;
    auto pdma = [&](int c, int buf) {
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

    // ---- row combination of Winograd row wi for the three patch columns of column pair jp.
    // row i of B^T d:  i = 0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
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
                // volatile: keeps hipcc from fusing the reads of a row into ds_read2_b64, which the LDS serves at a quarter of the rate of two
                // ds_read_b64 (MI355X_MICROARCH.md, LDS table: 16 vs 2 + 2 cycles per wavefront)
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
    // V(wi, 2 jp + jj) of the tile block, split and packed two input channels at a time: vf[term].u[k] = channels 2 k, 2 k + 1
    auto vmake = [&](WbFrag (&vf)[3], int jj) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = 2 * k + e;
                v[e] = jj ? __builtin_fmaf(beta, wY[q], wZ[q]) : wX[q] - wZ[q];
            }
            if (F16) sf_split2(v[0], v[1], SF_LOW_SCALE, vf[0].u[k], vf[1].u[k]);
            else {
                unsigned h[2], m[2], l[2];
                wb_split3(v[0], h[0], m[0], l[0]); wb_split3(v[1], h[1], m[1], l[1]);
                vf[0].u[k] = wb_pack(h[0], h[1]);
                vf[1].u[k] = wb_pack(m[0], m[1]);
                vf[2].u[k] = wb_pack(l[0], l[1]);
            }
        }
    };

    // ---- filter fragments of positions (wi, 2 jp), (wi, 2 jp + 1): [jj][mb][term], 12 consecutive fragments of the packed layout.
    // Buffer loads: the fragment index is wave-uniform (scalar offset), the only vector address is lane * 16.
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)upk, 0, (int)(ncg * nks * WB_FRAGS_PER_KSTEP * 1024), WB_RSRC_FLAGS);
    const unsigned fbase = (unsigned)((cg * nks * WB_FRAGS_PER_KSTEP + wi * 24 + jp * 12) * 1024);
    const unsigned lane16 = (unsigned)lane * 16u;
    WbFrag F[2][2][3];
    auto aload = [&](int c, int jj) {
        const unsigned so = fbase + (unsigned)((c * WB_FRAGS_PER_KSTEP + jj * 6) * 1024);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                F[jj][mb][t].q = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsF, lane16, so + (unsigned)((mb * 3 + t) * 1024), 0));
    };
    // partial products, small terms first (ta: filter term, tb: V term); the two channel blocks alternate
#define WB8_PROD(jj, nb, vf, ta, tb) do { \
        if (F16) { acc[jj][0][nb] = SF_MFMA(F[jj][0][ta].q, vf[tb].q, acc[jj][0][nb]); acc[jj][1][nb] = SF_MFMA(F[jj][1][ta].q, vf[tb].q, acc[jj][1][nb]); } \
        else     { acc[jj][0][nb] = WB_MFMA_BF(F[jj][0][ta].v, vf[tb].v, acc[jj][0][nb]); acc[jj][1][nb] = WB_MFMA_BF(F[jj][1][ta].v, vf[tb].v, acc[jj][1][nb]); } } while (0)
#define WB8_PHASE(jj, nb, vf) do { \
        if (F16) { WB8_PROD(jj, nb, vf, 2, 1); WB8_PROD(jj, nb, vf, 1, 0); WB8_PROD(jj, nb, vf, 0, 0); }      /* uq vl, ul vh, uh vh */ \
        else     { WB8_PROD(jj, nb, vf, 1, 1); WB8_PROD(jj, nb, vf, 0, 2); WB8_PROD(jj, nb, vf, 2, 0); \
                   WB8_PROD(jj, nb, vf, 0, 1); WB8_PROD(jj, nb, vf, 1, 0); WB8_PROD(jj, nb, vf, 0, 0); } } while (0)
#define WB8_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))      /* vmcnt(n) only; the builtin (not inline asm) so that hipcc's own wait bookkeeping sees it */

    // ---- prologue
    WB_STAMP(0);
    aload(0, 0); aload(0, 1);
    pdma(0, 0);
    WB8_VMCNT(0);
    WB_STAMP(1);
    pfix(0);
    __syncthreads();
    WB_STAMP(2);
    f32x16 acc[2][2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jj][mb][nb][r] = 0.f;

    // ---- K loop, software-pipelined.  A wavefront that alternates "produce V" and "its MFMAs" is blocked at the MFMA issue while the
    // matrix pipe drains -- and with the barrier its SIMD partner is in the same phase at the same time, so nothing overlaps.
    // tools/ubench/mfma_valu_bf16.hip: five VALU instructions per MFMA are free when they sit BETWEEN the MFMAs in program order.  So V is
    // double-buffered (vfA / vfB) and every block of MFMAs is written together with the production of the NEXT block's V:
    //     S1  products (jj 0, nb 0; vfA)   +  V(jj 1, nb 0) -> vfB
    //     S2  products (jj 1, nb 0; vfB)   +  row combinations of tile block 1, V(jj 0, nb 1) -> vfA
    //     S3  products (jj 0, nb 1; vfA)   +  V(jj 1, nb 1) -> vfB;   fragments (jj 0) of the next step requested
    //     -- vmcnt: the next step's patches are in; border fix-up; barrier --
    //     S4  products (jj 1, nb 1; vfB)   +  row combinations of tile block 0 of the NEXT step, V(jj 0, nb 0) -> vfA;
    //         fragments (jj 1) of the next step requested
    // Top of a step: vmcnt(0) (the step's fragments are in; hipcc would otherwise wait vmcnt(0) at their first use, i.e. for
    // the DMAs it cannot see), then the patch DMAs of the step after go out: three blocks of time to land.
    WbFrag vfA[3], vfB[3];
    wread(0, 0);
    vmake(vfA, 0);
    int c = 0;
    WB_STAMP(3);
    for (; c + 1 < nks; ++c) {
        WB8_VMCNT(0);
        pdma(c + 1, (c + 1) & 1);
        vmake(vfB, 1);
        WB8_PHASE(0, 0, vfA);
        wread(c & 1, 1);
        vmake(vfA, 0);
        WB8_PHASE(1, 0, vfB);
        vmake(vfB, 1);
        WB8_PHASE(0, 1, vfA);
        aload(c + 1, 0);
        WB_STAMP(4 + 3 * (c & 3));
        WB8_VMCNT(6);                                                     // the four DMAs are older than the six fragment loads
        pfix((c + 1) & 1);
        WB_STAMP(5 + 3 * (c & 3));
        __syncthreads();
        WB_STAMP(6 + 3 * (c & 3));
        wread((c + 1) & 1, 0);
        vmake(vfA, 0);
        WB8_PHASE(1, 1, vfB);
        aload(c + 1, 1);
    }
    WB_STAMP(16);
    WB8_VMCNT(0);
    vmake(vfB, 1);
    WB8_PHASE(0, 0, vfA);
    wread(c & 1, 1);
    vmake(vfA, 0);
    WB8_PHASE(1, 0, vfB);
    vmake(vfB, 1);
    WB8_PHASE(0, 1, vfA);
    WB8_PHASE(1, 1, vfB);
    WB_STAMP(17);
#undef WB8_PHASE
#undef WB8_PROD
#undef WB8_VMCNT

    // ---- output transform.  Row partials over j: pa = M0 + M1 + M2, pb = M1 - M2 - M3; wavefront (wi, 0) contributes (M0 + M1, M1),
    // wavefront (wi, 1) contributes (M2, -(M2 + M3)).  Round mb: part[row][jp][ab][nb][r4][lane] (float4 = registers 4 r4 .. 4 r4 + 3),
    // 8 x 16 KB; wavefront q then finishes tile block q & 1, register group q >> 1 (four channels) of the round.
    const int qnb = w & 1, qr4 = w >> 1;
    const int tr = 2 * qnb + tysub;
    const int ty = 4 * by + tr, tx = 16 * bx + col;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const size_t cstride = (size_t)Ho * Wo;
    float4 *part = (float4 *)lds;
    float gchk = 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        __syncthreads();                                    // patch stages (mb = 0) / the previous round's partials (mb = 1) are dead
        WB_STAMP(18 + 4 * mb);
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
        float bv[4], os[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bv[k] = bias ? bias[min(co0 + k, Cout - 1)] : 0.f;
            os[k] = F16 ? oscale[co0 + k] : 1.0f;                   // (padded to ncg * 64 entries)
        }
        WB_STAMP(19 + 4 * mb);
        __syncthreads();
        WB_STAMP(20 + 4 * mb);
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
            // A^T applied along i:  Y[0][.] = P0 + P1 + P2,  Y[1][.] = P1 - P2 - P3
            const float y0 = (WB_EL(P[0][0]) + WB_EL(P[1][0])) + WB_EL(P[2][0]), y1 = (WB_EL(P[0][1]) + WB_EL(P[1][1])) + WB_EL(P[2][1]);
            const float y2 = (WB_EL(P[1][0]) - WB_EL(P[2][0])) - WB_EL(P[3][0]), y3 = (WB_EL(P[1][1]) - WB_EL(P[2][1])) - WB_EL(P[3][1]);
#undef WB_EL
            if (F16 && guard) { MFR_GUARD_ACC(gchk, y0); MFR_GUARD_ACC(gchk, y1); MFR_GUARD_ACC(gchk, y2); MFR_GUARD_ACC(gchk, y3); }   // range guard (guard.h), before bias / residual / activation
            if (F16) { Y[k][0] = __builtin_fmaf(y0, os[k], bv[k]); Y[k][1] = __builtin_fmaf(y1, os[k], bv[k]); Y[k][2] = __builtin_fmaf(y2, os[k], bv[k]); Y[k][3] = __builtin_fmaf(y3, os[k], bv[k]); }
            else     { Y[k][0] = y0 + bv[k]; Y[k][1] = y1 + bv[k]; Y[k][2] = y2 + bv[k]; Y[k][3] = y3 + bv[k]; }
        }
        WB_STAMP(21 + 4 * mb);
        float *yb = y + ((size_t)b * Cout + co0) * cstride;
        const bool allco = cg * 64 + mb * 32 + 32 <= Cout;
        if (POOL) {
            float m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = fmaxf(fmaxf(Y[k][0], Y[k][1]), fmaxf(Y[k][2], Y[k][3]));     // the activations are monotone: act(max) = max(act)
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
            const bool interior = allco && r0 && r1 && c1 && !(W & 1);     // per lane; the common case: two float2 stores per channel
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
    if (F16) mfr_guard_commit(guard, gchk);
    WB_STAMP(26);
}

// ---- round 5: SuperPoint's conv1a FUSED into conv1b (f16x2) -------------------------------------------------------------------------------------
// conv1a (1 -> 64 channels) writes 6.4 GB per 64 images that conv1b reads straight back: 1.25 ms of HBM-bound writing, and in conv1b 17 % of a
// workgroup's time waiting for its first patches plus 0.8-1.7 k cycles per K step for the next ones (profiles/r05_conv1b_timeline.json).  Here a
// conv1b workgroup makes its own input: the 12 x 38 gray window of its 8 x 32 output block goes to LDS (one 4-byte load per thread), every wavefront
// evaluates conv1a + ReLU for eight of the 64 channels on the 10 x 34 patch (weights wave-uniform in scalar registers, a lane = one patch row x six
// columns, its 3 x 8 gray taps in registers for all eight channels) -- the SAME nine fused multiply-adds in the same order, + bias, max 0, as
// conv3x3_c1_relu_kernel (elementwise.hip), so the patch is that kernel's output bit for bit; positions outside the image are conv1b's zero padding
// -- and all four K steps' patches (120 KB) sit in LDS before the first MFMA.  The K loop then has no DMA, no border fix-up and no barrier: the
// wavefronts run free.  Everything from the row combinations to the output transform is the kernel's above (Cin = Cout = 64, pooled, ReLU).
__global__ void __launch_bounds__(512, 2) wino_split_c1_kernel(
    const float *__restrict__ gray, const float *__restrict__ w1a, const float *__restrict__ b1a, const uint4 *__restrict__ upk, const float *__restrict__ oscale,
    const float *__restrict__ bias, float *__restrict__ y, int H, int W, int nbx, int nby, int S, int Sx, int *guard)
{
    float gchk = 0.f;                                       // range guard (guard.h), over all blocks of this persistent workgroup
    constexpr int GS = 40;                                  // gray window row stride (38 columns used)
    __shared__ __attribute__((aligned(16))) float lds[32768 + 12 * GS];
    float *gwin = lds + 32768;                              // [12][GS] BEHIND the 128 KB the patch stages / output rounds use: it lives across blocks
    // PERSISTENT: one workgroup per CU walks the spatial blocks of its XCD's contiguous raster range (block = xcd Sx + slot + k per_xcd); the gray
    // window of the NEXT block (one value per thread) is requested before this block's patch is computed and lands during the block, so only the
    // first block of a workgroup pays the memory latency in front of its patch (3.5 k of a block's 30 k cycles, profiles/r05_conv1ab_timeline_nonpersistent.json)
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, per_xcd = gridDim.x >> 3;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w >> 1, jp = w & 1;
    auto gray_of = [&](int sl_) -> float {                  // this thread's element (tid < 456) of the 12 x 38 gray window: rows 8 by - 2 .. 8 by + 9, columns 32 bx - 2 .. 32 bx + 35; outside the image: 0 (conv1a's zero padding)
        const int s_ = xcd * Sx + sl_;
        if (tid >= 12 * 38 || sl_ >= Sx || s_ >= S) return 0.f;
        const int gr = tid / 38, gc = tid - 38 * gr;
        const int bx_ = s_ % nbx, by_ = (s_ / nbx) % nby, b_ = s_ / (nbx * nby);
        const int iy = 8 * by_ - 2 + gr, ix = 32 * bx_ - 2 + gc;
        return (iy >= 0 && iy < H && ix >= 0 && ix < W) ? gray[((size_t)b_ * H + iy) * W + ix] : 0.f;
    };
    // the window is double-staged: LDS holds the CURRENT block's, a register per thread the NEXT block's (requested a whole block ahead and written
    // to LDS right after the current patch is done -- not at the top of the next block, where the wait for it would also drain this block's
    // output stores: gfx950 counts loads and stores in one in-order counter)
    if (tid < 12 * 38) gwin[(tid / 38) * GS + (tid - 38 * (tid / 38))] = gray_of(slot);
    float gnext = gray_of(slot + per_xcd);
    for (int sl = slot; sl < Sx; sl += per_xcd) {
    const int s = xcd * Sx + sl;
    if (s >= S) break;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    // (the lane-derived offsets are re-derived per block from an opaque copy of the thread index: hoisted out of the block loop they are two
    // dozen loop-invariant registers that the K loop has no room for -- hipcc spilled them to scratch)
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63;
    const int col = lane & 15, tysub = (lane >> 4) & 1, kg = lane >> 5;

    // ---- filter fragments (requested below, once the gray window is in LDS: they land while the patch is computed)
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void *)upk, 0, (int)(4 * WB_FRAGS_PER_KSTEP * 1024), WB_RSRC_FLAGS);
    const unsigned fbase = (unsigned)((wi * 24 + jp * 12) * 1024);
    const unsigned lane16 = (unsigned)lane * 16u;
    WbFrag F[2][2][3];
    auto aload = [&](int c, int jj) {
        const unsigned so = fbase + (unsigned)((c * WB_FRAGS_PER_KSTEP + jj * 6) * 1024);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                F[jj][mb][t].q = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsF, lane16, so + (unsigned)((mb * 3 + t) * 1024), 0));
    };
    WB_STAMP(0);

    // ---- this block's gray window is in LDS (written during the previous block); the patch stages are free once every wavefront is past
    // the previous block's last output round
    __syncthreads();
    aload(0, 0); aload(0, 1);
    WB_STAMP(1);
    // ---- conv1a + ReLU on the patch: lane = (patch row pr < 10, column segment ps < 6): patch columns 6 ps .. 6 ps + 5 (34, 35 are never read)
    {
        const int pr = lane / 6, ps = lane - 6 * pr;
        if (pr < 10) {
            // the lane's 3 x 8 gray taps as PAIRS, once aligned on even and once on odd columns: the nine multiply-adds of two adjacent outputs are
            // one v_pk_fma_f32 each (no MFMA runs beside this phase, so packed fp32 is worth its two passes: half the instructions); every output
            // still sees its nine fmas in conv3x3_c1_relu_kernel's order
            wb_f32x2 ge[3][4], go[3][3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float2 v = *(const float2 *)(gwin + (pr + dy) * GS + 6 * ps + 2 * q); ge[dy][q] = wb_f32x2{v.x, v.y}; }
#pragma unroll
                for (int q = 0; q < 3; ++q) go[dy][q] = wb_f32x2{ge[dy][q].y, ge[dy][q + 1].x};
            }
            // patch position (row 8 by - 1 + pr, column 32 bx - 1 + 6 ps + j) inside the image?  outside: conv1b's zero padding
            const int py = 8 * by - 1 + pr, px0 = 32 * bx - 1 + 6 * ps;
            wb_f32x2 msk[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                msk[j].x = (py >= 0 && py < H && px0 + 2 * j >= 0 && px0 + 2 * j < W) ? 1.0f : 0.0f;
                msk[j].y = (py >= 0 && py < H && px0 + 2 * j + 1 >= 0 && px0 + 2 * j + 1 < W) ? 1.0f : 0.0f;
            }
            const bool all_in = __all(msk[0].x * msk[0].y * msk[1].x * msk[1].y * msk[2].x * msk[2].y != 0.0f);      // wave-uniform: interior workgroups skip the masking
#pragma unroll 1
            for (int t = 0; t < 8; ++t) {
                const int c = __builtin_amdgcn_readfirstlane(w + 8 * t);
                const float *k = w1a + 9 * c;
                float kk[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) kk[q] = k[q];
                const float bv = b1a[c];
                wb_f32x2 o[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    wb_f32x2 a = { 0.f, 0.f };
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        a = __builtin_elementwise_fma(wb_f32x2{kk[3 * dy], kk[3 * dy]}, ge[dy][j], a);
                        a = __builtin_elementwise_fma(wb_f32x2{kk[3 * dy + 1], kk[3 * dy + 1]}, go[dy][j], a);
                        a = __builtin_elementwise_fma(wb_f32x2{kk[3 * dy + 2], kk[3 * dy + 2]}, ge[dy][j + 1], a);
                    }
                    a = a + wb_f32x2{bv, bv};
                    o[j] = __builtin_elementwise_max(a, wb_f32x2{0.f, 0.f});
                }
                if (!all_in) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) o[j] = o[j] * msk[j];
                }
                float *dst = lds + c * WB_CH + pr * WB_RS + 6 * ps;
                *(float2 *)dst = make_float2(o[0].x, o[0].y); *(float2 *)(dst + 2) = make_float2(o[1].x, o[1].y); *(float2 *)(dst + 4) = make_float2(o[2].x, o[2].y);
            }
        }
    }
    WB_STAMP(2);
    __syncthreads();
    WB_STAMP(3);
    // every wavefront is done with this block's window: the next block's goes in, the one after that is requested
    if (tid < 12 * 38) gwin[(tid / 38) * GS + (tid - 38 * (tid / 38))] = gnext;
    gnext = gray_of(sl + 2 * per_xcd);

    // ---- from here on: wino_split_p8_kernel<true, true> with the K step's stage = its 16 channels in place (no DMA, no fix-up, no barrier)
    const int ra = (wi == 0) ? 0 : (wi == 2) ? 2 : 1;
    const int rb = (wi == 0) ? 2 : (wi == 2) ? 1 : (wi == 1) ? 2 : 3;
    const float sg = (wi == 1) ? 1.0f : -1.0f;
    const float beta = jp ? -1.0f : 1.0f;
    const int cX = jp ? 2 : 0, cZ = jp ? 1 : 2;
    const int ofa = (2 * tysub + ra) * WB_RS + 2 * col, ofb = (2 * tysub + rb) * WB_RS + 2 * col;
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
                a2[e] = *(const volatile wb_lds_f32x2 *)(ch + ofa + cX); a1[e] = *(const volatile wb_lds_f32 *)(ch + ofa + cZ);
                b2[e] = *(const volatile wb_lds_f32x2 *)(ch + ofb + cX); b1[e] = *(const volatile wb_lds_f32 *)(ch + ofb + cZ);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wX[e0 + e] = __builtin_fmaf(sg, b2[e].x, a2[e].x); wY[e0 + e] = __builtin_fmaf(sg, b2[e].y, a2[e].y);
                wZ[e0 + e] = __builtin_fmaf(sg, b1[e], a1[e]);
            }
            asm volatile("" : "+v"(wX[e0]), "+v"(wY[e0]), "+v"(wZ[e0]), "+v"(wX[e0 + 1]), "+v"(wY[e0 + 1]), "+v"(wZ[e0 + 1]),
                              "+v"(wX[e0 + 2]), "+v"(wY[e0 + 2]), "+v"(wZ[e0 + 2]), "+v"(wX[e0 + 3]), "+v"(wY[e0 + 3]), "+v"(wZ[e0 + 3]) :: "memory");
        }
    };
    auto vmake = [&](WbFrag (&vf)[3], int jj) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = 2 * k + e;
                v[e] = jj ? __builtin_fmaf(beta, wY[q], wZ[q]) : wX[q] - wZ[q];
            }
            sf_split2(v[0], v[1], SF_LOW_SCALE, vf[0].u[k], vf[1].u[k]);
        }
    };
    f32x16 acc[2][2][2];
#define WC1_PROD(jj, nb, vf, ta, tb) do { \
        acc[jj][0][nb] = SF_MFMA(F[jj][0][ta].q, vf[tb].q, acc[jj][0][nb]); acc[jj][1][nb] = SF_MFMA(F[jj][1][ta].q, vf[tb].q, acc[jj][1][nb]); } while (0)
#define WC1_PHASE(jj, nb, vf) do { WC1_PROD(jj, nb, vf, 2, 1); WC1_PROD(jj, nb, vf, 1, 0); WC1_PROD(jj, nb, vf, 0, 0); } while (0)
#define WC1_VMCNT0() __builtin_amdgcn_s_waitcnt(0x0F70)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jj][mb][nb][r] = 0.f;
    WbFrag vfA[3], vfB[3];
    wread(0, 0);
    vmake(vfA, 0);
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        WC1_VMCNT0();
        vmake(vfB, 1);
        WC1_PHASE(0, 0, vfA);
        wread(c, 1);
        vmake(vfA, 0);
        WC1_PHASE(1, 0, vfB);
        vmake(vfB, 1);
        WC1_PHASE(0, 1, vfA);
        aload(c + 1, 0);
        WB_STAMP(4 + 3 * c);
        wread(c + 1, 0);
        vmake(vfA, 0);
        WC1_PHASE(1, 1, vfB);
        aload(c + 1, 1);
    }
    WB_STAMP(16);
    WC1_VMCNT0();
    vmake(vfB, 1);
    WC1_PHASE(0, 0, vfA);
    wread(3, 1);
    vmake(vfA, 0);
    WC1_PHASE(1, 0, vfB);
    vmake(vfB, 1);
    WC1_PHASE(0, 1, vfA);
    WC1_PHASE(1, 1, vfB);
    WB_STAMP(17);
#undef WC1_VMCNT0
#undef WC1_PHASE
#undef WC1_PROD

    // ---- output transform, pooled + ReLU (as above)
    const int qnb = w & 1, qr4 = w >> 1;
    const int tr = 2 * qnb + tysub;
    const int ty = 4 * by + tr, tx = 16 * bx + col;
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t cstride = (size_t)Ho * Wo;
    float4 *part = (float4 *)lds;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        __syncthreads();
        WB_STAMP(18 + 4 * mb);
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
        const int co0 = mb * 32 + 4 * kg + 8 * qr4;
        float bv[4], os[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bv[k] = bias ? bias[co0 + k] : 0.f; os[k] = oscale[co0 + k]; }
        WB_STAMP(19 + 4 * mb);
        __syncthreads();
        WB_STAMP(20 + 4 * mb);
        const float4 *pq = part + (qnb * 4 + qr4) * 64 + lane;
        float4 P[4][2];
#pragma unroll
        for (int row = 0; row < 4; ++row)
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) {
                const float4 u = pq[(((row * 2 + 0) * 2 + ab) * 8) * 64], v = pq[(((row * 2 + 1) * 2 + ab) * 8) * 64];
                P[row][ab] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
            }
        float m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#define WB_EL(v) (k == 0 ? (v).x : k == 1 ? (v).y : k == 2 ? (v).z : (v).w)
            const float y0 = (WB_EL(P[0][0]) + WB_EL(P[1][0])) + WB_EL(P[2][0]), y1 = (WB_EL(P[0][1]) + WB_EL(P[1][1])) + WB_EL(P[2][1]);
            const float y2 = (WB_EL(P[1][0]) - WB_EL(P[2][0])) - WB_EL(P[3][0]), y3 = (WB_EL(P[1][1]) - WB_EL(P[2][1])) - WB_EL(P[3][1]);
#undef WB_EL
            if (guard) { MFR_GUARD_ACC(gchk, y0); MFR_GUARD_ACC(gchk, y1); MFR_GUARD_ACC(gchk, y2); MFR_GUARD_ACC(gchk, y3); }
            const float Y0 = __builtin_fmaf(y0, os[k], bv[k]), Y1 = __builtin_fmaf(y1, os[k], bv[k]), Y2 = __builtin_fmaf(y2, os[k], bv[k]), Y3 = __builtin_fmaf(y3, os[k], bv[k]);
            m[k] = fmaxf(fmaxf(fmaxf(Y0, Y1), fmaxf(Y2, Y3)), 0.f);
        }
        if (ty < Ho && tx < Wo) {
            float *yo = y + ((size_t)b * 64 + co0) * cstride + (size_t)ty * Wo + tx;
#pragma unroll
            for (int k = 0; k < 4; ++k) yo[(size_t)k * cstride] = m[k];
        }
        WB_STAMP(21 + 4 * mb);
    }
    WB_STAMP(26);
    }   // next block of this workgroup
    mfr_guard_commit(guard, gchk);
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------
static size_t wb_frag_bytes(int Cin, int Cout)
{
    const size_t ncg = (Cout + 63) / 64, nks = (Cin + 15) / 16;
    return ncg * nks * WB_FRAGS_PER_KSTEP * 1024;
}

template <bool F16>
static int wb_filter_transform(const float *w, int Cin, int Cout, void *upk, void *stream)
{
    if (!w || !upk || Cin <= 0 || Cout <= 0) return MFR_E_ARG;
    const size_t fb = wb_frag_bytes(Cin, Cout);
    const long long total = (long long)(fb / 16);
    float *oscale = F16 ? (float *)((char *)upk + fb) : nullptr;
    const int cpad = (Cout + 63) / 64 * 64;
    if (F16) hipLaunchKernelGGL(wb_filter_scale_kernel, dim3((unsigned)cpad), dim3(64), 0, (hipStream_t)stream, w, Cin, Cout, cpad, oscale);
    hipLaunchKernelGGL((wb_filter_kernel<F16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, (Cin + 15) / 16, total,
                       (const float *)oscale, (uint4 *)upk);
    CHECK_LAUNCH();
    return 0;
}

template <bool F16>
static int wb_conv(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W, int act, int pool, float *y, void *stream)
{
    if (!x || !upk || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || act < 0 || act > 2) return MFR_E_ARG;
    if (pool && (H < 2 || W < 2 || residual)) return MFR_E_ARG;
    if ((size_t)4 * Cin * H * W >= 0x7fffffffull) return MFR_E_ARG;    // one image must fit a 2 GB buffer descriptor
    const int nbx = ((W + 1) / 2 + 15) / 16, nby = ((H + 1) / 2 + 3) / 4;
    const int ncg = (Cout + 63) / 64, nks = (Cin + 15) / 16;
    const long long S = (long long)nbx * nby * B, Sx = (S + 7) / 8;
    const size_t fb = wb_frag_bytes(Cin, Cout);
    const int chunk = (fb > (size_t)(2u << 20)) ? 8 : 1;               // packed filters of all groups vs half an XCD's L2
    const long long grid = ((Sx + chunk - 1) / chunk) * (long long)chunk * ncg * 8;
    if (grid > 0x7fffffffll || fb >= 0x7fffffffull) return MFR_E_ARG;
    const float *oscale = F16 ? (const float *)((const char *)upk + fb) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (pool) hipLaunchKernelGGL((wino_split_p8_kernel<true, F16>), dim3((unsigned)grid), dim3(512), 0, st, x, (const uint4 *)upk, oscale, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg, nks, act, chunk, F16 ? mfr_guard_current() : (int *)nullptr);
    else      hipLaunchKernelGGL((wino_split_p8_kernel<false, F16>), dim3((unsigned)grid), dim3(512), 0, st, x, (const uint4 *)upk, oscale, bias, y, residual, Cin, Cout, H, W, nbx, nby, (int)S, (int)Sx, ncg, nks, act, chunk, F16 ? mfr_guard_current() : (int *)nullptr);
    CHECK_LAUNCH();
    return 0;
}

extern "C" {

#ifdef WB_PROF
int mfr_wino_split_profile(unsigned long long *out_host)      /* the stamps of the last launch: 8 wavefronts x 32 */
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(wb_prof), sizeof(unsigned long long) * 256, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : MFR_E_LAUNCH;
}
#endif

size_t mfr_wino_bf16x3_filter_bytes(int Cin, int Cout) { return (Cin <= 0 || Cout <= 0) ? 0 : wb_frag_bytes(Cin, Cout); }
size_t mfr_wino_f16x2_filter_bytes(int Cin, int Cout) { return (Cin <= 0 || Cout <= 0) ? 0 : wb_frag_bytes(Cin, Cout) + (size_t)((Cout + 63) / 64) * 64 * 4; }

int mfr_wino_bf16x3_filter_transform(const float *w, int Cin, int Cout, void *upk, void *stream) { return wb_filter_transform<false>(w, Cin, Cout, upk, stream); }
int mfr_wino_f16x2_filter_transform(const float *w, int Cin, int Cout, void *upk, void *stream) { return wb_filter_transform<true>(w, Cin, Cout, upk, stream); }

int mfr_conv3x3_wino_bf16x3(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                            int act, int pool, float *y, void *stream)
{
    return wb_conv<false>(x, upk, bias, residual, B, Cin, Cout, H, W, act, pool, y, stream);
}

int mfr_conv3x3_wino_f16x2(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                           int act, int pool, float *y, void *stream)
{
    return wb_conv<true>(x, upk, bias, residual, B, Cin, Cout, H, W, act, pool, y, stream);
}

int mfr_sp_conv1ab_f16x2(const float *gray, const float *w1a, const float *b1a, const void *upk1b, const float *bias1b, int B, int H, int W, float *y, void *stream)
{
    if (!gray || !w1a || !b1a || !upk1b || !y || B <= 0 || H < 2 || W < 2) return MFR_E_ARG;
    const int nbx = ((W + 1) / 2 + 15) / 16, nby = ((H + 1) / 2 + 3) / 4;
    const long long S = (long long)nbx * nby * B, Sx = (S + 7) / 8;
    if (S > 0x7fffffffll) return MFR_E_ARG;
    const long long grid = 8 * (Sx < 32 ? Sx : 32);                    // persistent: one workgroup (128 KB of LDS) per CU, 32 CUs per XCD
    const float *oscale = (const float *)((const char *)upk1b + wb_frag_bytes(64, 64));
    hipLaunchKernelGGL(wino_split_c1_kernel, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, gray, w1a, b1a, (const uint4 *)upk1b, oscale, bias1b, y,
                       H, W, nbx, nby, (int)S, (int)Sx, mfr_guard_current());
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
