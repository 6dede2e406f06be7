// split_f16.h -- the "f16x2" operand split of the matrix-core kernels (gemm_split.hip, winograd_split.hip, attention.hip).
//
// An fp32 product on the gfx950 F16 matrix cores, fp32 accumulate, THREE v_mfma_f32_32x32x16_f16 per K block instead of the six
// bf16 ones of the "bf16x3" split:
//
//     activation side   x  ->  xh = rne_f16(x),  xl = rne_f16((x - xh) * 2^11)          (two terms; x - xh is exact in fp32)
//     weight side       w' = w * s  (s = a power of two per OUTPUT feature, chosen at pack time so that max |w'| of the feature is in
//                       [2^14, 2^15) -- exact)   ->  wh = rne_f16(w'),  wl = rne_f16(w' - wh),  wq = rne_f16(wh * 2^-11)
//     product           x . w' ~= sum_k  xl wq + xh wl + xh wh        (small terms first, one fp32 accumulator; dropped: xl wl * 2^-11,
//                       below 2^-24 |x||w'|);  the epilogue multiplies by 1 / s.
//
// Why the low activation term is carried SCALED: (x - xh) is 2^-12 |x| or smaller, so unscaled it would leave the f16 normal range for every
// |x| < 0.25 and the pair would only be good to an ABSOLUTE 2^-25.  Scaled by 2^11 it has the exponent range of xh itself:
// xh + 2^-11 xl reproduces x to 2^-24 |x| (half an fp32 ulp -- the sign of the residual is the 24th bit) for 2^-14 <= |x| <= 65504, and
// to an absolute 2^-36 below that.  The 2^-11 is applied on the weight side where it costs nothing at run time (wq is a third packed
// term) and where the range is under control: with the per-feature scale, wl and wq are normal numbers for every weight within 2^-16 of
// the feature's largest one, and the smaller ones are represented to 2^-39 of it.
// Precondition: |x| <= 65504 (for the Winograd kernels: the transformed patch, i.e. |activation| < 16376); beyond it the f16 term is inf.
// Measured error class: tools/ubench/f16x2_probe.hip (profiles/r05_f16x2_probe.jsonl) next to the exact-fp32 MFMA and bf16x3.
//
// Cost: 2.5 VALU per activation element (bf16x3: 5.5) and half the MFMAs.
#pragma once
#include <hip/hip_runtime.h>

#define SF_LOW_SCALE 2048.0f

typedef _Float16 sf_f16x8 __attribute__((ext_vector_type(8)));

// (x0, x1) -> h = {rne_f16(x0), rne_f16(x1)},  l = {rne_f16((x0 - h0) * S), rne_f16((x1 - h1) * S)};  S = SF_LOW_SCALE, handed in so that it
// lives in ONE scalar register (VOP3P takes no literal)
__device__ __forceinline__ void sf_split2(float x0, float x1, float S, unsigned &h, unsigned &l)
{
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l) : "v"(r0), "s"(S));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l) : "v"(r1), "s"(S));
}

// weight side, one element (pack kernels only: plain conversions).  ws = w * s already applied by the caller.
__device__ __forceinline__ void sf_split_w(float ws, unsigned short &wh, unsigned short &wl, unsigned short &wq)
{
    const _Float16 h = (_Float16)ws;
    const _Float16 l = (_Float16)(ws - (float)h);
    const _Float16 q = (_Float16)((float)h * (1.0f / SF_LOW_SCALE));
    wh = __builtin_bit_cast(unsigned short, h);
    wl = __builtin_bit_cast(unsigned short, l);
    wq = __builtin_bit_cast(unsigned short, q);
}

// the per-feature power-of-two scale: max |w'| in [2^14, 2^15); an all-zero feature gets 1
__device__ __forceinline__ float sf_feature_scale(float maxabs)
{
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(maxabs, &e);                      // maxabs = m * 2^e, m in [0.5, 1)  ->  maxabs * 2^(15 - e) in [2^14, 2^15)
    int k = 15 - e;
    k = k > 126 ? 126 : (k < -126 ? -126 : k);
    return ldexpf(1.0f, k);
}

#define SF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf_f16x8, (a)), __builtin_bit_cast(sf_f16x8, (b)), (c), 0, 0, 0)
