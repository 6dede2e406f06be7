// conv_direct.hip -- 3x3 / stride 1 / pad 1 convolution (NCHW f32 in, f32 out) as a DIRECT implicit GEMM on the gfx950 f16 matrix cores at fp32
// accuracy (the f16x2 operand split, split_f16.h), with a halo tile staged ONCE in LDS in operand form and the nine taps as shifted reads of it.
// Same layer, same arguments, same epilogue (+bias, activation, optional residual, optional 2x2 max-pool) as mfr_conv3x3_wino_f16x2
// (winograd_split.hip); round 6, VERDICT r5 item 3.
//
// Reference call site: SuperGlue_matcher / LoFTR_matcher (etc/feature_matching_baselines/matchers.py:12-120) -> the un-vendored SuperPoint encoder
// (conv1b .. conv4b, convPa, convDa) and LoFTR ResNet-FPN backbone.
//
// Why a second kernel for the same layer.  The split-Winograd kernel executes 16 / 36 of the direct multiply-adds, but every transformed patch
// element costs ~4.5 vector instructions (transform + split) and is used for only 64 output channels (the 16 Winograd positions inflate the
// accumulators 4x, so a workgroup cannot hold more): 10-18 vector instructions per MFMA, matrix pipe 20-30 % busy (profiles/r05_pmc_*.json).  Here
// every input element is split ONCE per workgroup at staging and then serves 9 taps x 64 / 128 output channels straight from LDS: ~1.3 other
// instructions per MFMA, so the loop is bound by the matrix pipe itself (MI355X_MICROARCH.md: <= 5 single-issue instructions hide in an MFMA's
// 32 cycles).  2.25x the MFMAs at 2.5-3x the pipe utilisation.
//
// Mapping to CDNA4.
//   * workgroup = DC_NW = 4 wavefronts, TWO workgroups per CU (one wavefront of each per SIMD): a tile of TR rows x 32 columns of output pixels x 64 MG
//     output channels, MG in {1 (Cout <= 64), 2}: TR = 16 / 8.  Wavefront w: channel pair mg = w % MG (2 x 32 channels), row group ng = w / MG (4 rows = 4
//     pixel blocks of 32): 2 x 4 accumulator blocks of 32 x 32 = 128 registers (the last group of a 196-channel layer: 3 x 2, see `tail`).  A = weights
//     (32 channels x 16 input channels), B = pixels (16 input channels x 32 pixels of one row).
//   * K step = 16 input channels.  The (TR + 2) x 34 halo patch of the step is staged as [term (xh | xl)][channel half][pixel] x 16 bytes (eight
//     f16 channels of one pixel): a wavefront's B operand of tap (dy, dx) is ONE ds_read_b128 per lane at pixel (row + dy, column + dx), 32
//     consecutive 16-byte units per channel half -- conflict-free for every tap (SQ_LDS_BANK_CONFLICT = 0), no padding.  Two stages (81,920 / 49,152 B
//     of LDS per workgroup): the patch of step c + 1 is fetched (buffer loads, zeros outside the image and beyond Cin), split and written while step c
//     multiplies, spread over the 27 phases of the step; the K loop is ONE basic block with ONE barrier per K step (216 MFMAs per wavefront).
//   * weights: pre-scaled per output channel, split into (wh, wl, wq = wh 2^-11) and packed in operand order [64-channel group][K step][tap]
//     [block][term] x 1 KB; every wavefront streams its own 6 fragments per tap from L2 straight into registers TWO taps ahead (gfx950 returns
//     vector-memory loads in order: a shorter lead would make every weight wait a wait for the staging loads' HBM latency) -- no operand of the
//     weight side crosses LDS, nothing waits on a barrier.
//   * a tap = three phases of 8 independent MFMAs (wq.xl, wl.xh, wh.xh -- small terms first, one accumulator); the operand registers a phase frees
//     are reloaded in the next phase for the tap after (xl, xh from LDS) or the tap after that (weights).
//   * epilogue: every 32 x 32 accumulator block goes through 4 KB of wave-private LDS (lane = pixel -> lane = four consecutive pixels of one
//     channel) and leaves as four unconditional 16-byte buffer stores of eight full 128-byte lines; scale / bias / residual are fetched ahead of the
//     stores (one in-order counter for loads and stores).  Pooling: rows pair in registers before the exchange, columns pair inside a lane after it.
//   * grid: 1-D, XCD-aware: workgroup id & 7 = XCD; the channel groups of one spatial tile run back to back on the same XCD (its L2 holds the halo).
// What bounds it: the matrix pipe at the chip's power limit (62-74 % busy at 1.25-1.29 GHz, profiles/r06_pmc_dconv_*.json).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/mfr_hip.h"
#include "split_f16.h"
#include "guard.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

typedef float dc_f32x16 __attribute__((ext_vector_type(16)));
#define DC_RSRC_FLAGS 0x00020000
#define DC_OOB 0x80000000u
#define DC_PC 34                     // patch columns: 32 + halo
#define DC_FRAGS_PER_TAP 6           // 2 channel blocks x 3 terms
// wavefronts per workgroup: 4 = TWO independent workgroups per CU (one wavefront of each per SIMD).  A CU writes (and fetches) ~8 bytes per cycle
// whatever the instruction mix (profiles/r06_dconv_timeline_*.json: 256 KB of output per 8-wavefront tile = 27 k cycles of store tail in a 105 k
// cycle tile, and 6-12 k cycles before the first patch is staged): with one workgroup per CU all of that is exposed, with two the matrix pipe
// belongs to the other workgroup meanwhile.  8 (one workgroup per CU, twice the tile) is kept for the A/B: -DDC_NW=8.
#ifndef DC_NW
#define DC_NW 4
#endif

// Measurement build only (tools/dconv_timeline.py compiles THIS file a second time with -DDC_PROF into tools/ubench/libdconv_prof.so; the product
// library carries no instrumentation): s_memtime stamps of the eight wavefronts of one mid-grid workgroup.
#ifdef DC_PROF
__device__ unsigned long long dc_prof[8][64];
#define DC_STAMP(k) do { if (blockIdx.x == (gridDim.x / 2 | 5u) && lane == 0) { __builtin_amdgcn_sched_barrier(0); dc_prof[w][(k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
extern "C" int *mfr_guard_current(void) { return nullptr; }
#else
#define DC_STAMP(k) do { } while (0)
#endif

#define DC_IC(k) std::integral_constant<int, k>{}
// STR = 2 (round 6; LoFTR's strided 3x3 layers): the patch of a TR x 32 output tile is (2 TR + 1) x 65 input pixels, staged with its COLUMNS de-interleaved by
// parity -- row = [33 even columns | 33 odd columns] -- so that tap (dy, dx) of a row of 32 outputs is again 32 consecutive 16-byte units (row 2 r + dy,
// half dx & 1, column c + (dx >> 1)).  Four times the patch per output: MG = 4 (all four wavefronts share 4 rows, 256 channels per workgroup) keeps
// two stages within 80 KB.
template <int MG, int NW, int STR> struct DcGeom {
    static constexpr int NT = 64 * NW;                 // threads per workgroup
    static constexpr int NGW = NW / MG;                // row groups (4 rows each) per workgroup
    static constexpr int TR = 4 * NGW;                 // output rows per tile
    static constexpr int PR = STR == 1 ? TR + 2 : 2 * TR + 1;      // patch rows
    static constexpr int PCW = STR == 1 ? DC_PC : 66;  // staged row width (pixels)
    static constexpr int P = PR * PCW;                 // patch pixels
    static constexpr int PPAD = (P + 63) / 64 * 64;    // plane stride (16-byte units)
    static constexpr int STAGE = 4 * PPAD;             // [term][channel half][pixel]
    static constexpr int R = (2 * PPAD + NT - 1) / NT; // staging rounds: item = (channel half, pixel), NT per round
};

// ---- filters: w [Cout, Cin, 3, 3] f32 -> w' = w * s[co] split into (wh, wl, wq), packed as MFMA A operands: fragment
// f = ((((cg * nks + c) * 9 + tap) * 2 + m) * 3 + term), 64 lanes x 16 bytes; lane l holds output channel cg * 64 + m * 32 + (l & 31), input
// channels 16 c + 8 (l >> 5) + (0 .. 7); term 0 = wh, 1 = wl, 2 = wq.  Channels beyond Cin / Cout are zero.  The blob ends with the per-channel
// 1 / s (ncg x 64 floats, written by dc_scale_kernel BEFORE the pack kernel runs).
__global__ void __launch_bounds__(64) dc_scale_kernel(const float *__restrict__ w, int Cin, int Cout, int cpad, float *__restrict__ oscale)
{
    const int co = blockIdx.x, lane = threadIdx.x;
    float mx = 0.f;
    if (co < Cout)
        for (int t = lane; t < Cin * 9; t += 64) mx = fmaxf(mx, fabsf(w[(size_t)co * Cin * 9 + t]));
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0 && co < cpad) oscale[co] = 1.0f / sf_feature_scale(mx);
}
__global__ void __launch_bounds__(256) dc_pack_kernel(const float *__restrict__ w, int Cin, int Cout, int nks, long long total, const float *__restrict__ oscale, uint4 *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int l = (int)(t & 63);
    long long f = t >> 6;
    const int term = (int)(f % 3); f /= 3;
    const int m = (int)(f & 1); f >>= 1;
    const int tap = (int)(f % 9); f /= 9;
    const int c = (int)(f % nks);
    const int cg = (int)(f / nks);
    const int co = cg * 64 + m * 32 + (l & 31);
    const float s = 1.0f / oscale[co];
    unsigned short hw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = 16 * c + 8 * (l >> 5) + e;
        const float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] * s : 0.f;
        unsigned short wh, wl, wq;
        sf_split_w(v, wh, wl, wq);
        hw[e] = term == 0 ? wh : term == 1 ? wl : wq;
    }
    out[t] = make_uint4((unsigned)hw[0] | ((unsigned)hw[1] << 16), (unsigned)hw[2] | ((unsigned)hw[3] << 16),
                        (unsigned)hw[4] | ((unsigned)hw[5] << 16), (unsigned)hw[6] | ((unsigned)hw[7] << 16));
}

// ---- the convolution ----------------------------------------------------------------------------------------------------------------------------
template <int MG, int NW, bool POOL, int STR>
__global__ void __launch_bounds__(64 * NW, 8 / NW) conv_direct_f16x2_kernel(
    const float *__restrict__ x, const uint4 *__restrict__ wp, unsigned wp_bytes, const float *__restrict__ oscale, const float *__restrict__ bias,
    const float *__restrict__ residual, float *__restrict__ y, int Cin, int Cout, int H, int W, int nbx, int nby, int S, int ncgw, int nks, int act, int *guard, int ldrows)
{
    static_assert(STR == 1 || (STR == 2 && !POOL), "stride 1 or 2; no pooled strided variant");
    using G = DcGeom<MG, NW, STR>;
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * G::STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cgi = slot % ncgw, s = (slot / ncgw) * 8 + xcd;
    if (s >= S) return;
    const int bx = s % nbx, by = (s / nbx) % nby, b = s / (nbx * nby);
    const int x0 = 32 * bx, y0 = G::TR * by;            // (output coordinates)
    const int HW = H * W;
    const int Hout = STR == 1 ? H : (H - 1) / 2 + 1, Wout = STR == 1 ? W : (W - 1) / 2 + 1;
    // Wavefront blocking: MB channel blocks x NB pixel rows.  Standard (2 x 4): channel pair mg = w % MG, row group ng = w / MG of four rows.  TAIL (3 x 2;
    // LoFTR's 196-channel layers): the layer's last 128-channel group has only three blocks with real channels (128 .. 223 of 196): all four wavefronts
    // take the same three blocks and two rows each -- 6 accumulator blocks instead of 8, 12.5 % fewer MFMAs per such layer at equal work per wavefront
    // (a wavefront that merely skipped its dead block would idle at the step barriers while its SIMD's other workgroup may be in the same state).
    const bool tail = !POOL && NW == 4 && MG == 2 && cgi == ncgw - 1 && (Cout - cgi * 128 + 31) / 32 == 3;

    // ---- staging plan (K-loop invariant): round r, item i = NT r + tid -> channel half hh = i / PPAD (wave-uniform: PPAD % 64 == 0), patch pixel p
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)b * Cin * HW), 0, Cin * HW * 4, DC_RSRC_FLAGS);
    unsigned voff[G::R];
    int wdst[G::R], hh[G::R];
#pragma unroll
    for (int r = 0; r < G::R; ++r) {
        const int h = (G::NT * r + 64 * w) / G::PPAD;     // >= 2: this wavefront has no item in the (last, partial) round -- it stages zeros into the planes' padding
        const int p = G::NT * r + tid - h * G::PPAD;     // (no branch: the K loop stays ONE basic block and the scheduler spreads the staging over the MFMAs)
        const int pr = p / G::PCW, q = p - pr * G::PCW;
        const int pc = STR == 1 ? q : 2 * (q % 33) + q / 33;                        // patch column of the staged unit (STR 2: [even | odd] halves)
        const int gy = STR * y0 - 1 + pr, gx = STR * x0 - 1 + pc;
        const bool ok = h < 2 && p < G::P && pc < STR * 32 + 3 - STR && gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff[r] = ok ? (unsigned)(gy * W + gx) * 4u : DC_OOB;
        hh[r] = h < 2 ? h : 0;
        wdst[r] = h < 2 ? h * G::PPAD + p : G::P + (lane & 15);
    }
    float sv[8];
    auto sload = [&](int r, int c) {                     // the eight channels 16 c + 8 hh + (0 .. 7) of this thread's pixel; outside the image / beyond Cin: 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = 16 * c + 8 * hh[r] + e;
            sv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, ch < Cin ? voff[r] : DC_OOB, (unsigned)ch * (unsigned)HW * 4u, 0));
        }
    };
    auto swrite = [&](int r, int buf) {
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sf_split2(sv[2 * k], sv[2 * k + 1], SF_LOW_SCALE, h[k], l[k]);
        lds[buf * G::STAGE + wdst[r]] = make_uint4(h[0], h[1], h[2], h[3]);
        lds[buf * G::STAGE + 2 * G::PPAD + wdst[r]] = make_uint4(l[0], l[1], l[2], l[3]);
    };

    // ---- weight fragments: [m][term] of the current tap; fragment index wave-uniform (scalar offset), vector address = lane * 16
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, (int)wp_bytes, DC_RSRC_FLAGS);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto body = [&](auto mbc, auto nbc) {
    constexpr int MB = decltype(mbc)::value, NB = decltype(nbc)::value;
    const int mg = MB == 2 ? w % MG : 0, ng = MB == 2 ? w / MG : w;
    const int mb0 = (cgi * MG + mg) * 2;                 // this wavefront's first 32-channel block (global index)
    const unsigned kstride = (unsigned)nks * (9u * DC_FRAGS_PER_TAP * 1024u);       // bytes per 64-channel group
    auto aload = [&](int c, int tap, int m, int term) -> uint4 {        // tap may be 9, 10 (= taps 0, 1 of step c + 1)
        const int mbg = mb0 + m;
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsW, lane16, (unsigned)(mbg >> 1) * kstride + (unsigned)(((c * 9 + tap) * 2 + (mbg & 1)) * 3 + term) * 1024u, 0));
    };
    const int rdb = (lane >> 5) * G::PPAD + (STR * NB * ng) * G::PCW + (lane & 31);
    auto tapoff = [](int n, int dy, int dx) { return STR == 1 ? (n + dy) * G::PCW + dx : (2 * n + dy) * G::PCW + (dx & 1) * 33 + (dx >> 1); };

    DC_STAMP(0);
    // ---- prologue: stage 0 (all rounds' loads in flight together: the accumulators are not live yet), the first two taps' weights
    {
        float pv[G::R][8];
#pragma unroll
        for (int r = 0; r < G::R; ++r) { sload(r, 0);
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[r][e] = sv[e]; }
#pragma unroll
        for (int r = 0; r < G::R; ++r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[e] = pv[r][e];
            swrite(r, 0); }
    }
    // Weight registers: two sets [m][term] (term 0 wh, 1 wl, 2 wq), E for the even taps of a step, O for the odd ones; a term's registers are reloaded
    // for the tap TWO ahead one phase after their last use: five phases (40 MFMAs, ~1300 cycles) between a fragment's request and its use.  gfx950
    // returns vector-memory loads in order, so a wait for a weight fragment also waits for every OLDER load -- the staging loads below are HBM
    // misses (~1 k cycles): with a one-tap lead every staging round would stall the matrix stream.
    uint4 AE[MB][3], AO[MB][3], Bh[NB], Bl[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) { AE[m][0] = aload(0, 0, m, 0); AE[m][1] = aload(0, 0, m, 1); AE[m][2] = aload(0, 0, m, 2); AO[m][1] = aload(0, 1, m, 1); AO[m][2] = aload(0, 1, m, 2); AO[m][0] = AE[m][0]; }
    dc_f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // staging of step c + 1 inside step c, in phase units (27 per step): round r is requested in phase SL(r) and split + written in phase SW(r) -- four
    // or more phases (~1 k cycles) later, ONE register set (the next round is requested after this one is written)
#define DC_SL(r) (G::R == 5 ? 5 * (r) : 8 * (r))
#define DC_SW(r) (G::R == 5 ? ((r) == 4 ? 26 : 5 * (r) + 4) : ((r) == 2 ? 24 : 8 * (r) + 6))
    DC_STAMP(1);
    for (int c = 0; c < nks; ++c) {
        DC_STAMP(2 + 4 * (c & 7));
        __syncthreads();                                 // stage c complete (written during step c - 1 / the prologue); stage c - 1's buffer is free
        const uint4 *st = lds + (c & 1) * G::STAGE + rdb;
        const int nbuf = (c + 1) & 1;
#pragma unroll
        for (int n = 0; n < NB; ++n) Bl[n] = st[2 * G::PPAD + tapoff(n, 0, 0)];
        DC_STAMP(3 + 4 * (c & 7));
        auto stage_event = [&](int phi) {
#pragma unroll
            for (int r = 0; r < G::R; ++r) {             // (the last step stages a step of zeros: channels >= Cin, no memory traffic)
                if (phi == DC_SW(r)) swrite(r, nbuf);
                if (phi == DC_SL(r)) sload(r, c + 1);
            }
        };
        auto do_tap = [&](auto tapc, uint4 (&A)[MB][3], uint4 (&Ao)[MB][3]) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int dy = tap / 3, dx = tap - 3 * dy;
            constexpr int dyn = (tap + 1) / 3, dxn = (tap + 1) - 3 * dyn;
            // phase 1: wq . xl  |  requests: this tap's xh, the OTHER set's wh (tap + 1)
#pragma unroll
            for (int n = 0; n < NB; ++n) Bh[n] = st[tapoff(n, dy, dx)];
            __builtin_amdgcn_sched_barrier(0);           // the reads first: phase 2 needs them 8 MFMAs from here
#pragma unroll
            for (int m = 0; m < MB; ++m) Ao[m][0] = aload(c, tap + 1, m, 0);
            stage_event(3 * tap);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = SF_MFMA(A[m][2], Bl[n], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: wl . xh  |  requests: wq of tap + 2, the next tap's xl
#pragma unroll
            for (int m = 0; m < MB; ++m) A[m][2] = aload(c, tap + 2, m, 2);
            if (tap < 8) {
#pragma unroll
                for (int n = 0; n < NB; ++n) Bl[n] = st[2 * G::PPAD + tapoff(n, dyn, dxn)];
            }
            stage_event(3 * tap + 1);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = SF_MFMA(A[m][1], Bh[n], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
            // phase 3: wh . xh  |  requests: wl of tap + 2
#pragma unroll
            for (int m = 0; m < MB; ++m) A[m][1] = aload(c, tap + 2, m, 1);
            stage_event(3 * tap + 2);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = SF_MFMA(A[m][0], Bh[n], acc[m][n]);
            __builtin_amdgcn_sched_barrier(0);
        };
        do_tap(DC_IC(0), AE, AO); do_tap(DC_IC(1), AO, AE); do_tap(DC_IC(2), AE, AO); DC_STAMP(4 + 4 * (c & 7)); do_tap(DC_IC(3), AO, AE); do_tap(DC_IC(4), AE, AO);
        do_tap(DC_IC(5), AO, AE); DC_STAMP(5 + 4 * (c & 7)); do_tap(DC_IC(6), AE, AO); do_tap(DC_IC(7), AO, AE); do_tap(DC_IC(8), AE, AO);
        // nine taps: the sets change roles (O holds tap 0 of the next step, E's wl / wq its tap 1)
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) { const uint4 tmp = AE[m][t]; AE[m][t] = AO[m][t]; AO[m][t] = tmp; }
    }
#undef DC_SL
#undef DC_SW

    const int H = Hout, W = Wout, HW = Hout * Wout;      // from here on: the OUTPUT image (they shadow the input's; the staging lambdas above keep the input's)
    // ---- epilogue: lane = pixel (row y0 + 4 ng + n, column x0 + (lane & 31)), register r of block m = channel cg64 * 64 + 32 m + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    DC_STAMP(34);
    float gchk = 0.f;
    if (guard) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) MFR_GUARD_ACC(gchk, acc[m][n][r]);      // before bias / residual / activation (guard.h)
        mfr_guard_commit(guard, gchk);
    }
    // The store tail.  Measured on the first versions (profiles/r06_dconv_timeline_*.json): (1) gfx950 returns loads AND stores through one in-order
    // counter, so a vector load that is waited for behind a store waits for the store's acknowledgement (~1 k cycles): 32 x [load scale, bias; wait;
    // 4 stores] cost 40-80 k cycles per tile; (2) a predicated plain store costs a compare, two exec-mask updates and a branch: 30 cycles per store
    // instruction per CU, 16 as an unconditional buffer store whose offset lies beyond the buffer where nothing must be written; (3) the rest is per
    // INSTRUCTION, not per byte.  Hence: scale / bias are fetched once, before any store; every store is an unconditional buffer store; and where the
    // image width allows it (W % 4 == 0) each 32 x 32 accumulator block goes through 4 KB of wave-private LDS (lane = pixel -> lane = four consecutive
    // pixels of one channel) and leaves as FOUR 16-byte stores of eight full 128-byte lines instead of sixteen 4-byte ones.
    const int px = x0 + (lane & 31);
    const int half = lane >> 5;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const size_t cstride = (size_t)Ho * Wo;
    const int cpad1 = (Cout + 63) / 64 * 64 - 1;
    auto activate = [&](float v) { return act == 1 ? fmaxf(v, 0.f) : act == 2 ? (v > 0.f ? v : 0.01f * v) : v; };
    // invalid channel part 0x40000000, invalid pixel part 0x80000000: any sum of the two lies beyond a buffer of < 2^30 bytes (host check), none wraps
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void *)(y + (size_t)b * Cout * cstride), 0, (int)(Cout * cstride * 4), DC_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)(residual ? residual + (size_t)b * Cout * HW : nullptr), 0, residual ? Cout * HW * 4 : 0, DC_RSRC_FLAGS);   // (one image)
    if (ldrows > 0) {
        // ROWS output (round 6): y [B H W, ldrows] token-major, channel c of pixel p at y[p * ldrows + c] -- what the linear layer behind the convolution
        // reads (SuperPoint's descriptor head, LoFTR's fine map): the accumulators already hold four consecutive channels of a pixel per register
        // group, so each group leaves as one 16-byte store and the NCHW -> rows transposition pass (mfr_nchw_to_rows) disappears.
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void *)(y + (size_t)b * HW * ldrows), 0, HW * ldrows * 4, DC_RSRC_FLAGS);
        unsigned pixrow[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int oy = y0 + NB * ng + n;
            pixrow[n] = (oy < H && px < W) ? (unsigned)(oy * W + px) * (unsigned)ldrows * 4u : DC_OOB;
        }
        DC_STAMP(36);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (mb0 + m) * 32 + 8 * q + 4 * half;                     // four consecutive channels (Cout % 4 == 0: all in or all out)
                float o4[4], b4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { o4[k] = oscale[min(c0 + k, cpad1)]; b4[k] = bias ? bias[min(c0 + k, Cout - 1)] : 0.f; }
                const unsigned cpart = c0 < Cout ? (unsigned)c0 * 4u : 0x40000000u;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    typedef unsigned dc_u32x4 __attribute__((ext_vector_type(4)));
                    const float4 v = make_float4(activate(__builtin_fmaf(acc[m][n][4 * q], o4[0], b4[0])), activate(__builtin_fmaf(acc[m][n][4 * q + 1], o4[1], b4[1])),
                                                 activate(__builtin_fmaf(acc[m][n][4 * q + 2], o4[2], b4[2])), activate(__builtin_fmaf(acc[m][n][4 * q + 3], o4[3], b4[3])));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dc_u32x4, v), rsQ, cpart + pixrow[n], 0, 0);
                }
            }
    } else if (!(W & 3)) {
        const int L8 = lane >> 3, L7 = lane & 7;
        float os[MB][4], bv[MB][4];
        unsigned cho[MB][4];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = (mb0 + m) * 32 + 8 * j + L8;                       // this lane's channel of read-back j
                os[m][j] = oscale[min(co, cpad1)];                                    // (padded to ncg * 64 entries)
                bv[m][j] = bias ? bias[min(co, Cout - 1)] : 0.f;
                cho[m][j] = co < Cout ? (unsigned)co * (unsigned)cstride * 4u : 0x40000000u;
            }
        unsigned pix[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int oy = y0 + NB * ng + n, xg = x0 + 4 * L7;
            if (POOL) pix[n] = ((oy >> 1) < Ho && (xg >> 1) < Wo) ? (unsigned)((oy >> 1) * Wo + (xg >> 1)) * 4u : DC_OOB;      // (n even; W % 4 == 0: a pair is in or out as a whole)
            else      pix[n] = (oy < H && xg < W) ? (unsigned)(oy * W + xg) * 4u : DC_OOB;
        }
        __syncthreads();                                 // every wavefront is through with both stages (its last step's staging writes included)
        DC_STAMP(36);
        float *tb = (float *)lds + 1024 * w;             // wave-private 32 channels x 32 pixels
        constexpr int NSTEP = POOL ? 2 : 1;
        constexpr int DC_RD = 6;                         // residual requests in flight ahead of the stores
        float4 rv[POOL ? 1 : MB * NB * 4];
        auto rload = [&](int it) {                       // it = (m * NB + n) * 4 + j
            if (!POOL) rv[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsR, cho[it / (4 * NB)][it & 3] + pix[(it >> 2) % NB], 0, 0));
        };
        if (!POOL && residual) {
#pragma unroll
            for (int it = 0; it < DC_RD; ++it) rload(it);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; n += NSTEP) {
                if (m == 1 && n == 0) DC_STAMP(37);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tb[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + (lane & 31)] = POOL ? fmaxf(acc[m][n][r], acc[m][n + 1][r]) : acc[m][n][r];   // (scale > 0: max commutes with the epilogue)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = *(const float4 *)(tb + (8 * j + L8) * 32 + 4 * L7);
                    if (POOL) {
                        const float v0 = activate(__builtin_fmaf(fmaxf(t.x, t.y), os[m][j], bv[m][j])), v1 = activate(__builtin_fmaf(fmaxf(t.z, t.w), os[m][j], bv[m][j]));
                        typedef unsigned dc_u32x2 __attribute__((ext_vector_type(2)));
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(dc_u32x2, make_float2(v0, v1)), rsY, cho[m][j] + pix[n], 0, 0);
                    } else {
                        const int it = (m * NB + n) * 4 + j;
                        float4 v = make_float4(__builtin_fmaf(t.x, os[m][j], bv[m][j]), __builtin_fmaf(t.y, os[m][j], bv[m][j]),
                                               __builtin_fmaf(t.z, os[m][j], bv[m][j]), __builtin_fmaf(t.w, os[m][j], bv[m][j]));
                        if (residual) {
                            if (it + DC_RD < MB * NB * 4) rload(it + DC_RD);
                            v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w;
                        }
                        v = make_float4(activate(v.x), activate(v.y), activate(v.z), activate(v.w));
                        typedef unsigned dc_u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dc_u32x4, v), rsY, cho[m][j] + pix[n], 0, 0);
                    }
                }
            }
    } else {
        // any width: lane = pixel, register = channel, 4-byte stores (128-byte runs along x per channel)
        float osv[MB][16], bvv[MB][16];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (mb0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                osv[m][r] = oscale[min(co, cpad1)];
                bvv[m][r] = bias ? bias[min(co, Cout - 1)] : 0.f;
            }
        unsigned pixoff[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int oy = y0 + NB * ng + n;
            if (POOL) pixoff[n] = (!(lane & 1) && (oy >> 1) < Ho && (px >> 1) < Wo) ? (unsigned)((oy >> 1) * Wo + (px >> 1)) * 4u : DC_OOB;      // (n even)
            else      pixoff[n] = (oy < H && px < W) ? (unsigned)(oy * W + px) * 4u : DC_OOB;
        }
        auto choff = [&](int m, int r) -> unsigned {
            const int co = (mb0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            return co < Cout ? (unsigned)co * (unsigned)cstride * 4u : 0x40000000u;
        };
        DC_STAMP(36);
        if (POOL) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned co4 = choff(m, r);
#pragma unroll
                    for (int n = 0; n < NB; n += 2) {
                        float v = fmaxf(__builtin_fmaf(acc[m][n][r], osv[m][r], bvv[m][r]), __builtin_fmaf(acc[m][n + 1][r], osv[m][r], bvv[m][r]));
                        v = activate(fmaxf(v, __shfl_xor(v, 1)));                       // the activations are monotone: act(max) = max(act)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, co4 + pixoff[n], 0, 0);
                    }
                }
        } else {
            constexpr int DC_RD = 4;                                                     // iterations (of four loads) in flight ahead of the stores
            float rv[MB * 16][NB];
            auto rload = [&](int it) {
                const unsigned co4 = choff(it >> 4, it & 15);
#pragma unroll
                for (int n = 0; n < NB; ++n) rv[it][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, co4 + pixoff[n], 0, 0));
            };
            if (residual) {
#pragma unroll
                for (int it = 0; it < DC_RD; ++it) rload(it);
            }
#pragma unroll
            for (int it = 0; it < MB * 16; ++it) {
                const int m = it >> 4, r = it & 15;
                const unsigned co4 = choff(m, r);
                if (residual && it + DC_RD < MB * 16) rload(it + DC_RD);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    float v = __builtin_fmaf(acc[m][n][r], osv[m][r], bvv[m][r]);
                    if (residual) v += rv[it][n];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, activate(v)), rsY, co4 + pixoff[n], 0, 0);
                }
            }
        }
    }
    };
    if (tail) body(DC_IC(3), DC_IC(2));
    else body(DC_IC(2), DC_IC(4));
    DC_STAMP(35);
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------
static size_t dc_frag_bytes(int Cin, int Cout)
{
    const size_t ncg = (Cout + 63) / 64, nks = (Cin + 15) / 16;
    return ncg * nks * 9 * DC_FRAGS_PER_TAP * 1024;
}

extern "C" {

#ifdef DC_PROF
int mfr_dconv_occupancy(int mg, int pool)                /* workgroups per CU the runtime grants the instantiation */
{
    int n = -1;
    if (mg == 1 && !pool) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_direct_f16x2_kernel<1, DC_NW, false, 1>, 64 * DC_NW, 0);
    if (mg == 1 && pool)  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_direct_f16x2_kernel<1, DC_NW, true, 1>, 64 * DC_NW, 0);
    if (mg == 2 && !pool) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_direct_f16x2_kernel<2, DC_NW, false, 1>, 64 * DC_NW, 0);
    if (mg == 2 && pool)  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_direct_f16x2_kernel<2, DC_NW, true, 1>, 64 * DC_NW, 0);
    return n;
}
int mfr_dconv_profile(unsigned long long *out_host)      /* the stamps of the last launch: 8 wavefronts x 64 */
{
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(dc_prof), sizeof(unsigned long long) * 512, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : MFR_E_LAUNCH;
}
#endif

size_t mfr_conv3x3_direct_f16x2_filter_bytes(int Cin, int Cout) { return (Cin <= 0 || Cout <= 0) ? 0 : dc_frag_bytes(Cin, Cout) + (size_t)((Cout + 63) / 64) * 64 * 4; }

int mfr_conv3x3_direct_f16x2_filter_pack(const float *w, int Cin, int Cout, void *packed, void *stream)
{
    if (!w || !packed || Cin <= 0 || Cout <= 0) return MFR_E_ARG;
    const size_t fb = dc_frag_bytes(Cin, Cout);
    const long long total = (long long)(fb / 16);
    float *oscale = (float *)((char *)packed + fb);
    const int cpad = (Cout + 63) / 64 * 64;
    hipLaunchKernelGGL(dc_scale_kernel, dim3((unsigned)cpad), dim3(64), 0, (hipStream_t)stream, w, Cin, Cout, cpad, oscale);
    hipLaunchKernelGGL(dc_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, (Cin + 15) / 16, total, (const float *)oscale, (uint4 *)packed);
    CHECK_LAUNCH();
    return 0;
}

static int dc_conv(const float *x, const void *packed, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                   int act, int pool, float *y, int ldrows, void *stream)
{
    if (!x || !packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || act < 0 || act > 2) return MFR_E_ARG;
    if (pool && (H < 2 || W < 2 || residual)) return MFR_E_ARG;
    if ((size_t)4 * (Cin + 15) * H * W >= 0x7fffffffull || (size_t)4 * Cout * H * W >= 0x40000000ull) return MFR_E_ARG;     // one image (with the K padding) must fit a 2 GB buffer descriptor
    const size_t fb = dc_frag_bytes(Cin, Cout);
    if (fb >= 0x7fffffffull) return MFR_E_ARG;
    const int nks = (Cin + 15) / 16;
    const float *oscale = (const float *)((const char *)packed + fb);
    const int mg = Cout > 64 ? 2 : 1;
    const int tr = 4 * DC_NW / mg;
    const int nbx = (W + 31) / 32, nby = (H + tr - 1) / tr, ncgw = (Cout + 64 * mg - 1) / (64 * mg);
    const long long S = (long long)nbx * nby * B;
    const long long grid = ((S + 7) / 8) * 8 * ncgw;
    if (grid > 0x7fffffffll) return MFR_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int *guard = mfr_guard_current();
#define DC_LAUNCH(MGV, POOLV) hipLaunchKernelGGL((conv_direct_f16x2_kernel<MGV, DC_NW, POOLV, 1>), dim3((unsigned)grid), dim3(64 * DC_NW), 0, st, x, (const uint4 *)packed, (unsigned)fb, oscale, bias, \
                                                  residual, y, Cin, Cout, H, W, nbx, nby, (int)S, ncgw, nks, act, guard, ldrows)
    if (mg == 1) { if (pool) DC_LAUNCH(1, true); else DC_LAUNCH(1, false); }
    else         { if (pool) DC_LAUNCH(2, true); else DC_LAUNCH(2, false); }
#undef DC_LAUNCH
    CHECK_LAUNCH();
    return 0;
}

int mfr_conv3x3_direct_f16x2(const float *x, const void *packed, const float *bias, const float *residual, int B, int Cin, int Cout, int H, int W,
                             int act, int pool, float *y, void *stream)
{
    return dc_conv(x, packed, bias, residual, B, Cin, Cout, H, W, act, pool, y, 0, stream);
}

int mfr_conv3x3_direct_f16x2_rows(const float *x, const void *packed, const float *bias, int B, int Cin, int Cout, int H, int W, int act, float *yrows, int ldy, void *stream)
{
    if (ldy < Cout || (ldy & 3) || (Cout & 3) || H <= 0 || W <= 0 || (size_t)4 * ldy * H * W >= 0x40000000ull) return MFR_E_ARG;
    return dc_conv(x, packed, bias, nullptr, B, Cin, Cout, H, W, act, 0, yrows, ldy, stream);
}

int mfr_conv3x3s2_direct_f16x2(const float *x, const void *packed, const float *bias, int B, int Cin, int Cout, int H, int W, int act, float *y, void *stream)
{
    if (!x || !packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || act < 0 || act > 2) return MFR_E_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((size_t)4 * (Cin + 15) * H * W >= 0x7fffffffull || (size_t)4 * Cout * Ho * Wo >= 0x40000000ull) return MFR_E_ARG;
    const size_t fb = dc_frag_bytes(Cin, Cout);
    if (fb >= 0x7fffffffull) return MFR_E_ARG;
    const int nks = (Cin + 15) / 16;
    const float *oscale = (const float *)((const char *)packed + fb);
    const int nbx = (Wo + 31) / 32, nby = (Ho + 3) / 4, ncgw = (Cout + 255) / 256;
    const long long S = (long long)nbx * nby * B;
    const long long grid = ((S + 7) / 8) * 8 * ncgw;
    if (grid > 0x7fffffffll) return MFR_E_ARG;
    hipLaunchKernelGGL((conv_direct_f16x2_kernel<4, 4, false, 2>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (const uint4 *)packed, (unsigned)fb, oscale, bias,
                       (const float *)nullptr, y, Cin, Cout, H, W, nbx, nby, (int)S, ncgw, nks, act, mfr_guard_current(), 0);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
