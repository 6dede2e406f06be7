// kabsch.hip -- differentiable least-squares rotation of the regression heads (batched Kabsch), forward and backward, with NO
// host synchronisation.
//
// Reference: `procrustes` (lib/utils/solver.py:4-37), called from the Procrustes heads (lib/models/regression/head.py:55-163):
//     H = A_c^T B_c,  U S V^T = svd(H),  R = V diag(1, 1, det(V U^T)) U^T
// torch.linalg.svd synchronises the host (it reads the solver's `info` back), which is the one thing that keeps the whole
// training step from being captured as a HIP graph.  The rotation itself does not need an SVD:
//   forward   Horn's closed form: R is the rotation of the eigenvector of the largest eigenvalue of the symmetric 4x4 matrix
//             N(H) (unit quaternion), found by a fixed number of Jacobi sweeps -- the proper rotation maximising tr(R H), i.e.
//             exactly the reflection-corrected Kabsch solution;
//   backward  with M = H^T = R S (S = R^T M symmetric, eigen-decomposition S = U diag(s) U^T, one s negative in the reflection
//             case):  dR = R W_dM,  (W S + S W) = R^T dM - dM^T R.  For an incoming G = dL/dR, with A = skew(R^T G) and
//             W'_ij = (U^T A U)_ij / (s_i + s_j):   dL/dM = 2 R (U W' U^T),  dL/dH = (dL/dM)^T.
// One thread per matrix (the batch is the training batch: 10-32 matrices), binary64 inside, f32 at the boundary.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfr_hip.h"

#define CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return MFR_E_LAUNCH; } while (0)

#define KB_FN static __device__
#include "kabsch_math.h"

__global__ void __launch_bounds__(64) kabsch_fwd_kernel(const float *__restrict__ H, int B, float *__restrict__ R)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double S[9], Rd[9];
    for (int i = 0; i < 9; ++i) S[i] = (double)H[(size_t)b * 9 + i];
    kb_horn_rotation(S, Rd);
    for (int i = 0; i < 9; ++i) R[(size_t)b * 9 + i] = (float)Rd[i];
}

// the forward's rotation is recomputed in binary64 from H (not taken from its f32 rounding)
__global__ void __launch_bounds__(64) kabsch_bwd_kernel(const float *__restrict__ H, const float *__restrict__ G, int B, float *__restrict__ gH)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    double Hd[9], Gd[9], gd[9];
    for (int i = 0; i < 9; ++i) { Hd[i] = (double)H[(size_t)b * 9 + i]; Gd[i] = (double)G[(size_t)b * 9 + i]; }
    kb_rotation_backward(Hd, Gd, gd);
    for (int i = 0; i < 9; ++i) gH[(size_t)b * 9 + i] = (float)gd[i];
}

extern "C" {

// H [B, 3, 3] row-major (H = A_c^T B_c) -> R [B, 3, 3]: the proper rotation with B_c ~ A_c R^T
int mfr_kabsch_fwd(const float *H, int B, float *R, void *stream)
{
    if (!H || !R || B <= 0) return MFR_E_ARG;
    hipLaunchKernelGGL(kabsch_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, H, B, R);
    CHECK_LAUNCH();
    return 0;
}

// gR = dL/dR [B, 3, 3] -> gH = dL/dH [B, 3, 3]
int mfr_kabsch_bwd(const float *H, const float *gR, int B, float *gH, void *stream)
{
    if (!H || !gR || !gH || B <= 0) return MFR_E_ARG;
    hipLaunchKernelGGL(kabsch_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, H, gR, B, gH);
    CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
