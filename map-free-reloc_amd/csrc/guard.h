// guard.h -- the f16x2 RANGE GUARD (round 6).
//
// The f16x2 arithmetic (split_f16.h) carries an fp32 activation x as xh = rne_f16(x), xl = rne_f16((x - xh) 2^11): for |x| > 65504 (Winograd
// kernels: the TRANSFORMED patch value, i.e. |activation| >= 16376) xh is +-inf and xl is -+inf (or NaN).  Every product sum that contains such an
// element then holds inf - inf, inf * 0 or NaN: the fp32 ACCUMULATOR of every output the element contributes to is NaN or +-inf -- never a finite
// number -- whatever the weights are (w = 0 gives inf * 0 = NaN).  Activations, not accumulators, are what a ReLU / max / softmax can launder
// (v_max_f32 drops a NaN operand), so each f16x2 kernel tests its accumulators BEFORE the activation: chk = fma(acc, 0, chk) turns NaN for a NaN or
// infinite accumulator and stays 0 otherwise; one v_cmp + ballot per wavefront at the end, one atomicOr per offending wavefront.  The test therefore
// covers every INPUT of every f16x2 kernel, whoever produced it (an own kernel, a torch op, the caller), at < 1 % of the kernel's instructions
// (32-64 fused multiply-adds per thread and output tile), and it also fires for non-finite inputs.
//
// The flag is a device int the CALLER owns (a torch tensor per pipeline object); it reaches the kernels through ONE piece of per-thread
// library state, set by mfr_f16x2_guard_bind (include/mfr_hip.h): the f16x2 entry points keep their signatures.  Unbound (the default): no test.
#pragma once
#include <hip/hip_runtime.h>

#ifdef __cplusplus
extern "C" {
#endif
int *mfr_guard_current(void);          // the calling thread's bound flag (device pointer) or NULL; defined in elementwise.hip
#ifdef __cplusplus
}
#endif

#define MFR_GUARD_ACC(chk, v) ((chk) = __builtin_fmaf((v), 0.0f, (chk)))

__device__ __forceinline__ void mfr_guard_commit(int *guard, float chk)
{
    if (guard && __builtin_amdgcn_ballot_w64(chk != chk) != 0ull) {
        if ((threadIdx.x & 63) == 0) atomicOr(guard, 1);
    }
}
