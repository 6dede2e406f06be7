/*
 * include/mfr_hip.h -- C-ABI of libmfr_hip.so, the MI355X (gfx950) drop-in for the
 * feature-matching + scale-from-depth relative-pose hot path of
 * nianticlabs/map-free-reloc.
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes; no torch types.  All data pointers are
 *     DEVICE pointers unless the parameter name ends in _host.
 *   - the caller allocates every buffer (including the workspace, whose size is
 *     returned by the matching *_workspace_bytes query); the library keeps no
 *     per-call state and is re-entrant per stream.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - return value: 0 = launched OK, <0 = argument / launch error (MFR_E_*).
 *     Per-pair failures are VALUES, like the reference's NaN-pose convention
 *     (pose_solver.py:30-33,188-189,197-198,230-233): status[b] != 0 and R,t filled
 *     with NaN, n_inliers = 0.  A failing pair never aborts the batch.
 *   - batches: leading dim B = image pairs; correspondences are fixed-stride
 *     [B, maxN, 2] float32 with a count n_corr[B] (the device-side twin of the
 *     NaN-padded [Npairs, maxN, 4] npz wire format, utils.py:59-69).
 *   - images are H x W row-major float32.  Intrinsics K are [B,3,3] row-major with zero skew and
 *     bottom row [0,0,1], handed over IN THE DTYPE THE `data` DICT HOLDS THEM, tagged by `k_dtype`
 *     (one tag for K0 and K1):
 *       MFR_K_F64  float64 -- the Map-free loader's flow: correct_intrinsic_scale multiplies a
 *                  float64 np.eye(3) into K (lib/datasets/utils.py:117-130; always called,
 *                  lib/datasets/mapfree.py:50-52, config/mapfree.yaml:7-8), so np.linalg.inv(K)
 *                  (pose_solver.py:16), the K-normalisation (:39-40) and the threshold mean (:43)
 *                  are float64 arithmetic in the reference, and so they are here;
 *       MFR_K_F32  float32 -- resize=None datasets (K stays as parsed): the same three steps are
 *                  float32 arithmetic and only their results are promoted (quirk Q5).
 *     Both flows are pinned bit-for-bit against the reference's own Python
 *     (tests/golden/ref_k64.npz, ref_backproject.npz, ref_pnp_lift.npz, ref_emat_metric.npz).
 *
 * Every entry point names the reference interface it replaces (file:line under
 * the upstream repository).
 */
#ifndef MFR_HIP_H
#define MFR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFR_ABI_VERSION 6   /* 6 (round 6): mfr_conv3x3_direct_f16x2*, mfr_conv3x3s2_direct_f16x2, mfr_mlp_ln_*; 5 (round 6): mfr_f16x2_guard_bind (the f16x2 range guard); 2: intrinsics as (const void *K, int k_dtype) instead of const float *; 3: mfr_emat_solve_batch takes the
                             * model-quality method (MAGSAC++ / count) and its table; 4 (round 5): the f16x2 entry points (mfr_gemm_f16x2*,
                             * mfr_wino_f16x2_*, mfr_conv3x3_wino_f16x2, mfr_conv_igemm_f16x2), mfr_sg_attention_variant renumbered (0 f16x2,
                             * 1 exact fp32, 2 bf16x3), the measurement-only entry points (mfr_conv3x3_wino_bf16x3_variant,
                             * mfr_wino_bf16x3_profile, the GEMM ablation flags) removed */

/* intrinsics dtype tags */
#define MFR_K_F32 0
#define MFR_K_F64 1

/* library-level error codes (negative) */
#define MFR_E_ARG      (-1)
#define MFR_E_LAUNCH   (-2)
#define MFR_E_WORKSPACE (-3)

/* per-pair status values (same as oracle/mfr_oracle.h) */
#define MFR_ST_OK          0
#define MFR_ST_TOO_FEW     1   /* fewer correspondences than the solver minimum (Q9) */
#define MFR_ST_BAD_DEPTH   2   /* too few correspondences with valid depth          */
#define MFR_ST_NO_MODEL    3   /* RANSAC produced no model / refinement failed      */
#define MFR_ST_DEGENERATE  4   /* |t| > 1000 (pose_solver.py:223-225)               */

int mfr_abi_version(void);

/* ---- f16x2 range guard (round 6).  The reference's networks are plain fp32 (etc/feature_matching_baselines/matchers.py:50,105 call fp32 PyTorch
 *      modules: no input-range precondition); the f16x2 kernels (mfr_gemm_f16x2*, mfr_conv_igemm_f16x2, mfr_conv3x3_wino_f16x2, mfr_sp_conv1ab_f16x2,
 *      mfr_sg_attention variant 0) represent an activation as two f16 terms and need |x| <= 65504 (Winograd: |x| < 16376).  After
 *      mfr_f16x2_guard_bind(flag) every such launch issued by THIS host thread ORs 1 into *flag (a device int the caller owns and clears) when one of
 *      its fp32 accumulators is NaN / +-inf before the activation -- which is the case for every output that an out-of-range (or non-finite) input
 *      element contributes to (csrc/guard.h).  flag = NULL unbinds (the default: no test).  The bound pointer is the ONE piece of state the library
 *      keeps (thread-local); it is read when a launch is issued, so a captured HIP graph keeps the pointer it was captured with.
 *      Host side: pipeline.py folds the flag into the per-pair status (MFR_ST_RANGE) and re-runs such a batch in the exact bf16x3 arithmetic. ---- */
int mfr_f16x2_guard_bind(int *device_flag);
#define MFR_ST_RANGE 7         /* host-side status (pipeline.py): the f16x2 range guard fired for this batch; no pose unless re-run in bf16x3 */
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char *mfr_target_arch(void);

/* ---- numerics self-test hook: out[i] = {a/b, sqrt(a), a*b+c (unfused)} in f64, used by
 *      tests to prove the host/device IEEE contract the bit-exact parity relies on ---- */
int mfr_test_f64_ops(const double *a, const double *b, const double *c, int n, double *out3, void *stream);
/* raw RNG / sampler exposure for parity tests (oracle: mfr_ref_sample_distinct) */
int mfr_test_sample(uint64_t seed, const int64_t *pair_ids, int B, int iters, int n, int k, int32_t *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * PnP path:  PnPSolver.estimate_pose, lib/models/matching/pose_solver.py:184-235
 *   int-truncate pts0 (:186) -> depth gather (:192-193) -> valid = d > depth.min() (:196, Q6)
 *   -> backproject_3d with K0 (:206, :6-17) -> cv.solvePnPRansac(P3P, iters, thr, conf) (:209-213)
 *   -> refit + cv.solvePnPGeneric(ITERATIVE) when >= 6 inliers (:216-220) -> |t| > 1000 reject.
 * Outputs: R [B,9] f64 row-major, t [B,3] f64, n_inliers [B] i32 (= len(inliers), the
 * submission confidence, model.py:37), status [B] i32, inlier_mask [B,maxN] u8 indexed by
 * the ORIGINAL correspondence index (may be NULL).
 * RNG: Philox4x32-10 keyed by (seed, pair_ids[b], iteration) -- the reference has no seed
 * knob (OpenCV fixed RNG); results are bit-reproducible for fixed (seed, pair_id).
 * ------------------------------------------------------------------------------------------ */
size_t mfr_pnp_workspace_bytes(int B, int maxN, int max_iters);
int mfr_pnp_solve_batch(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                        const float *depth0, int H, int W,
                        const void *K0, const void *K1, int k_dtype,
                        int max_iters, double reproj_thr, double confidence,
                        uint64_t seed, const int64_t *pair_ids,
                        void *workspace, size_t workspace_bytes,
                        double *R, double *t, int32_t *n_inliers, int32_t *status, uint8_t *inlier_mask,
                        void *stream);

/* stage-level entry points of the PnP path (same arithmetic, exposed for parity tests and
 * for callers that already hold lifted 3-D points).  xyz [B,maxN,3] f64, obs [B,maxN,2] f64,
 * src_idx [B,maxN] i32, n_valid [B] i32. */
int mfr_depth_min(const float *depth, int B, int H, int W, float *partial_min /*[B,16]*/, void *stream);
int mfr_pnp_lift(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                 const float *depth0, const float *partial_min, int H, int W, const void *K0, int k_dtype,
                 double *xyz, double *obs, int32_t *src_idx, int32_t *n_valid, void *stream);
int mfr_pnp_ransac(const double *xyz, const double *obs, const int32_t *n_valid, int B, int maxN,
                   const void *K1, int k_dtype, int max_iters, double reproj_thr, double confidence,
                   uint64_t seed, const int64_t *pair_ids,
                   int32_t *counts /*[B,max_iters] workspace*/, int32_t *inl_idx /*[B,maxN] workspace*/,
                   double *R, double *t, int32_t *n_inliers, int32_t *status,
                   uint8_t *mask_valid /*[B,maxN] over lifted points, may be NULL*/,
                   int32_t *best_iter /*[B] may be NULL*/, int32_t *iters_run /*[B] may be NULL*/,
                   void *stream);

/* ------------------------------------------------------------------------------------------
 * Metric scale from depth: EssentialMatrixMetricSolver.estimate_pose, pose_solver.py:137-172
 *   mask == 1 inliers (:137) -> int-truncate both views (:138-139) -> depth gather (:140-141)
 *   -> valid = d0 > 0 & d1 > 0 (:144) -> back-project (:150-151) -> xyz0 <- R xyz0 (:154)
 *   -> scale_i = (xyz1_i - xyz0_i) . t (:157) -> exhaustive 1-D RANSAC, first strict max
 *   (:160-166, Q2) -> t_metric = best_scale * t (:169).
 * R [B,9], t [B,3] f64 inputs (unit translation); outputs t_metric [B,3] f64 (NaN on failure),
 * best_scale [B] f64, n_inliers [B] i32 (scale-consensus count = submission confidence),
 * status [B] (in_status != 0 is passed through: "inliers == 0 -> return", :131-132;
 * no valid depth -> MFR_ST_BAD_DEPTH, :145-149).
 * emat_mask [B,maxN] u8 may be NULL (= all correspondences).
 * ------------------------------------------------------------------------------------------ */
size_t mfr_scale_workspace_bytes(int B, int maxN);
int mfr_scale_from_depth_batch(const float *pts0, const float *pts1, const uint8_t *emat_mask,
                               const int32_t *n_corr, int B, int maxN,
                               const float *depth0, const float *depth1, int H, int W,
                               const void *K0, const void *K1, int k_dtype,
                               const double *R, const double *t,
                               const int32_t *in_status /* [B] status of the E-mat stage, may be NULL */,
                               double scale_thr,
                               void *workspace, size_t workspace_bytes,
                               double *t_metric, double *best_scale, int32_t *n_inliers, int32_t *status,
                               void *stream);

/* ------------------------------------------------------------------------------------------
 * Essential-matrix path: EssentialMatrixSolver.estimate_pose, lib/models/matching/pose_solver.py:29-61
 *   K-normalise in K's dtype (:39-40) -> thr = pix_thr / mean(fx0, fy1, fy0, fx1) (:43, Q8)
 *   -> cv.findEssentialMat(USAC_MAGSAC, prob) (:46-48): 5-point RANSAC, max_iters (OpenCV default 1000; the reference does
 *      not override it), adaptive iteration cap driven by the number of points under thr, and -- score_method
 *      MFR_EMAT_SCORE_MAGSAC, the method the reference names -- MAGSAC++ model quality (sigma-marginalised loss, 4 degrees of
 *      freedom, k = 3.64, k sigma_max = max_thr_ratio * thr; smallest total loss wins) with sigma-consensus++ local optimisation
 *      (IRLS with the MAGSAC++ weights on (R, unit t): every new best model from iteration 100 on, and the winner at the end).
 *      MFR_EMAT_SCORE_COUNT: inlier count at thr + one LM polish of (R, t) (rounds 1-3; kept for A/B).
 *   -> cv.recoverPose per E (:56-60): 4 decompositions, cheirality vote.
 * magsac_lut: DEVICE copy of the table mfr_magsac_lut (host) fills, 2 * (lut_m + 1) doubles, lut_m <= 2048 (ignored for COUNT).
 * Outputs: R [B,9] (orthonormal to rounding), t [B,3] UNIT translation f64 (NaN on failure), n_inliers [B] = number of
 * cheirality-passing inliers (the `n` recoverPose returns), inlier_mask [B,maxN] u8 = those
 * inliers (what self.mask aliases after the loop, Q7; feed it to mfr_scale_from_depth_batch),
 * status [B].  best_iter / iters_run / counts_out [B,max_iters] / losses_out [B,max_iters] (MAGSAC) / lo_runs [B] (number of
 * local optimisations) are optional diagnostics (NULL ok).
 * ------------------------------------------------------------------------------------------ */
#define MFR_EMAT_SCORE_MAGSAC 0
#define MFR_EMAT_SCORE_COUNT  1
#define MFR_MAGSAC_LUT_M 2048
int mfr_magsac_lut(double *lut_host /* [2 * (M + 1)] */, int M);
size_t mfr_emat_workspace_bytes(int B, int maxN, int max_iters);
int mfr_emat_solve_batch(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                         const void *K0, const void *K1, int k_dtype, double pix_thr, double confidence, int max_iters,
                         uint64_t seed, const int64_t *pair_ids, int score_method, const double *magsac_lut, int lut_m,
                         double max_thr_ratio, void *workspace, size_t workspace_bytes,
                         double *R, double *t, int32_t *n_inliers, int32_t *status, uint8_t *inlier_mask,
                         int32_t *best_iter, int32_t *iters_run, int32_t *counts_out, double *losses_out, int32_t *lo_runs,
                         void *stream);

/* ------------------------------------------------------------------------------------------
 * Procrustes path: ProcrustesSolver.estimate_pose, lib/models/matching/pose_solver.py:238-320 with
 * PROCRUSTES.REFINE False (config/matching/mapfree/sg_procrustes_dptkitti.yaml; the optional ICP
 * refinement is mfr_procrustes_icp_refine below).  int-truncate both views (:248-249) -> depth gather (:256-258) -> valid vs each map's
 * minimum (:261, Q6) -> back-project both (:273-274) -> o3d registration_ransac_based_on_correspondence
 * (3-point Kabsch RANSAC, max_corr_dist, confidence 0.999, best = fitness then RMSE, final re-fit)
 * (:286-287) -> inliers = int(fitness * N) (:288).  max_iters caps Open3D's 100000.
 * No model found -> identity pose with 0 inliers and status OK (Open3D's default result), as upstream.
 * ------------------------------------------------------------------------------------------ */
size_t mfr_procrustes_workspace_bytes(int B, int maxN, int max_iters);
int mfr_procrustes_solve_batch(const float *pts0, const float *pts1, const int32_t *n_corr, int B, int maxN,
                               const float *depth0, const float *depth1, int H, int W, const void *K0, const void *K1, int k_dtype,
                               double max_corr_dist, double confidence, int max_iters, uint64_t seed, const int64_t *pair_ids,
                               void *workspace, size_t workspace_bytes, double *R, double *t, int32_t *n_inliers,
                               int32_t *status, int32_t *best_iter, int32_t *iters_run, int32_t *counts_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * SuperPoint post-processing.  Reference call site: SuperGlue_matcher.match,
 * etc/feature_matching_baselines/matchers.py:93-120 (hyper-parameters :65-71); the network itself
 * is the un-vendored magicleap submodule (.gitmodules:4-6), restated per SURVEY.md Appendix A.2.
 *   mfr_sp_scoremap            softmax-65, drop dustbin, 8x8 pixel shuffle: logits [B,65,Hc,Wc]
 *                              (NCHW) -> scores [B,8Hc,8Wc]
 *   mfr_sp_nms_candidates      simple_nms(radius 4, 2 rounds) + keypoint threshold + border
 *                              removal; appends survivors to cand [B,cand_cap] (u64 key = score
 *                              bits << 32 | ~raster index) and counts them; nms_out [B,H,W] optional
 *   mfr_sp_select_topk         top-K (K <= 1024) by (score desc, raster index asc); raster order
 *                              when <= K candidates -> kpts [B,K,2] (x,y), kscores [B,K], n_kpts [B]
 *   mfr_sp_sample_descriptors  dense descriptors [B,Hc,Wc,256] (NHWC, un-normalised) -> per-cell L2,
 *                              bilinear grid_sample(align_corners=True), L2 -> desc [B,K,256]
 * ------------------------------------------------------------------------------------------ */
int mfr_sp_scoremap(const float *logits, int B, int Hc, int Wc, float *scores, void *stream);
int mfr_sp_nms_candidates(const float *scores, int B, int H, int W, int nms_radius, float threshold, int border,
                          float *nms_out, uint64_t *cand, int cand_cap, int32_t *cand_count, void *stream);
int mfr_sp_select_topk(const uint64_t *cand, int cand_cap, const int32_t *cand_count, int B, int W, int K,
                       float *kpts, float *kscores, int32_t *n_kpts, void *stream);
int mfr_sp_sample_descriptors(const float *dense_nhwc, int B, int Hc, int Wc, const float *kpts,
                              const int32_t *n_kpts, int K, float *desc, void *stream);

/* ------------------------------------------------------------------------------------------
 * SuperGlue kernels (same call site; upstream algorithm per SURVEY.md Appendix A.3).
 *   mfr_sg_attention       softmax(q k^T / 8) v, `heads` heads x 64, fp32 in / fp32 out, scores never materialised.
 *                          Both contractions run on the 16-bit matrix cores at fp32 accuracy by operand splitting (csrc/attention.hip).
 *                          mfr_sg_attention_variant: 0 = f16x2 (round 5, what mfr_sg_attention runs; csrc/split_f16.h: every operand as two f16
 *                          terms, main + correction accumulators, three partial products; precondition |q|, |k|, |v| <= 65504),
 *                          2 = bf16x3 (rounds 3-4: exact 3-way bf16 split, six partial products), 1 = the exact-fp32 matrix-core kernel of
 *                          rounds 1-2 (v_mfma_f32_32x32x2_f32), kept for A/B and tests; error vs fp64 of all three = the fp32 class
 *                          (tests/test_gpu_nets_parity.py).  q,k,v [B2,N,ld] (head h = channels
 *                          [64h, 64h+64) from each base pointer), out [B2,N,ldo]; keys/queries
 *                          >= n_tok[image] are masked; cross != 0 -> image b reads K/V of image b^1.
 *   mfr_sg_sinkhorn_match  log_optimal_transport(S, bin_score, iters) without materialising the
 *                          dustbin-augmented matrix + mutual arg-max + exp(score) > match_thr +
 *                          ordered compaction (matchers.py:111-116) into pts0/pts1 [B,maxN,2],
 *                          n_corr [B] -- the layout mfr_pnp_solve_batch consumes.
 *                          S [B,ldS,ldS] = mdesc0^T mdesc1 / 16, n0/n1 [B] true keypoint counts,
 *                          kpts0/kpts1 [B,K,2]; also matches0 [B,ldS] (-1 = none), mscores0 [B,ldS].
 *                          mfr_sg_sinkhorn_match_variant: 0 = one sweep over S per iteration (round 4, default when ldS % 4 == 0 and
 *                          ldS <= 1024), 1 = a row pass and a column pass per iteration (rounds 1-3); the same bits either way.
 * ------------------------------------------------------------------------------------------ */
/*   mfr_gemm_f16x2 / mfr_gemm_bf16x3   the transformers' linear layers, y [M, ldy] (+)= act(x [M, ldx] W [N, K]^T + bias), fp32 in / fp32 out, on
 *                          the 16-bit matrix cores at fp32 accuracy by operand splitting (csrc/gemm_split.hip; rounds 1-2 called the library's fp32
 *                          GEMM here).  f16x2 (round 5, what nets/linear.py runs; csrc/split_f16.h): activations as two f16 terms, the weight
 *                          pre-scaled per output feature and packed as three f16 terms, three partial products, fp32 accumulate;
 *                          precondition |x| <= 65504.  bf16x3 (rounds 3-4): exact 3-way bf16 split of both operands, six partial products.
 *                          W is split and packed once per weight set (mfr_gemm_*_pack, size from mfr_gemm_*_pack_bytes; 0 if K % 32 != 0;
 *                          the two packed formats are not interchangeable).  flags: 1 = ReLU, 2 = accumulate into y (y += x W^T + bias).
 *                          x 16-byte aligned, ldx % 4 == 0; bias may be NULL.  Kernel selection for the bitwise-agreement test (every
 *                          kernel sums each output element in the same order): + 4 one tile per workgroup (round 3), + 8 persistent
 *                          workgroups with register-staged W; none = persistent, W by LDS-DMA (K % 64 == 0; other K run as + 8). */
size_t mfr_gemm_f16x2_pack_bytes(int N, int K);
int mfr_gemm_f16x2_pack(const float *w, int N, int K, void *packed, void *stream);
int mfr_gemm_f16x2(const float *x, int ldx, const void *packed_w, const float *bias, float *y, int ldy, int M, int N, int K, int flags, void *stream);
/*   mfr_gemm_f16x2_batched   nbatch products of one shape in one launch, y_b [M, ldy] = x_b [M, ldx] w_b [N, K]^T * out_mul: the matchers' score /
 *                          similarity matrices (SuperGlue's S = mdesc0 mdesc1^T / 16, models/superglue.py:278-279 of the submodule; LoFTR's
 *                          feat_c0 feat_c1^T / C, coarse_matching.py:104-106), which rounds 1-4 left to the library's batched fp32 GEMM.  The second
 *                          operand of each product is packed per launch into caller scratch (mfr_gemm_f16x2_pack_batched: nbatch blobs of
 *                          mfr_gemm_f16x2_pack_bytes(N, K) each, consecutive; row stride ldw and batch stride in elements), with out_mul -- a power
 *                          of two, so exact -- folded into the per-row scale; batch strides of x and y in elements (x's a multiple of 4). */
int mfr_gemm_f16x2_pack_batched(const float *w, int ldw, int nbatch, long long w_batch_stride, int N, int K, float out_mul, void *packed, void *stream);
int mfr_gemm_f16x2_batched(const float *x, int ldx, long long x_batch_stride, const void *packed_w, const float *bias, float *y, int ldy, long long y_batch_stride,
                           int nbatch, int M, int N, int K, int flags, void *stream);
/*   mfr_conv_igemm_f16x2   implicit-GEMM convolution on NCHW images in the f16x2 arithmetic (csrc/gemm_split.hip): the strided / 1x1 / 7x7
 *                          convolutions of the matcher backbones (LoFTR's conv1, the stride-2 3x3 and 1x1 of layer2.0 / layer3.0, the FPN's 1x1
 *                          lateral and output convolutions; rounds 1-4: MIOpen / hipBLASLt).  y [B,Cout,Ho,Wo] = act(conv(x [B,Cin,H,W], w, stride,
 *                          pad) + bias), Ho = (H + 2 pad - KH) / stride + 1; relu != 0: ReLU; bias may be NULL.  packed_w = mfr_gemm_f16x2_pack of
 *                          the [Cout, K] matrix with K = mfr_conv_igemm_k(Cin, KH, KW) and k = (dy KW + dx) * Cpad + ci, Cpad = Cin rounded up to 32
 *                          (zero columns for ci >= Cin); Cin == 1: k = dy KW + dx, zero columns up to K. */
int mfr_conv_igemm_k(int Cin, int KH, int KW);
int mfr_conv_igemm_f16x2(const float *x, const void *packed_w, const float *bias, float *y, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                         int stride, int pad, int relu, void *stream);
/*   mfr_conv_igemm_f16x2_upadd (round 6): the same convolution (no activation) + the FPN merge of LoFTR's backbone in one pass,
 *                          y = conv(x) + bias + F.interpolate(lo, scale_factor=2, mode='bilinear', align_corners=True) with lo [B,Cout,Hl,Wl], Ho = 2 Hl,
 *                          Wo = 2 Wl (upstream ResNetFPN_8_2.forward: x2_out = layer2_outconv(x2) + x3_out_2x, x1_out likewise; call site matchers.py:50);
 *                          the up-sampling arithmetic is mfr_upsample2x_add's. */
int mfr_conv_igemm_f16x2_upadd(const float *x, const void *packed_w, const float *bias, const float *lo, int Hl, int Wl, float *y, int B, int Cin, int H, int W,
                               int Cout, int KH, int KW, int stride, int pad, void *stream);
/*   mfr_gemm_f16x2_ln / mfr_gemm_bf16x3_ln (round 6): y = [y +] LayerNorm_N(x W^T + bias) * gamma + beta for N = 128 (one feature block) and K % 64 == 0:
 *                          the linear layer with the LayerNorm that follows it in upstream LoFTREncoderLayer.forward (`message = self.norm1(self.merge(...))`,
 *                          `message = self.norm2(self.mlp(...)); return x + message`; call site matchers.py:50) in ONE launch, for the fine-level encoder
 *                          (d_model 128) whose tensors (2.4 M rows) make every pass HBM-bound.  accumulate != 0: y += (the residual, in place).
 *                          Two-pass statistics (mean, then centred squares), biased variance, rsqrt(var + eps): the arithmetic of mfr_layernorm. */
int mfr_gemm_f16x2_ln(const float *x, int ldx, const void *packed_w, const float *bias, const float *gamma, const float *beta, float eps, float *y, int ldy,
                      int M, int N, int K, int accumulate, void *stream);
int mfr_gemm_bf16x3_ln(const float *x, int ldx, const void *packed_w, const float *bias, const float *gamma, const float *beta, float eps, float *y, int ldy,
                       int M, int N, int K, int accumulate, void *stream);
/*   mfr_gemm_f16x2_windows / mfr_gemm_bf16x3_windows (round 6): upstream FinePreprocess.forward (F.unfold(feat_f, 5x5, stride 4, padding 2) -> rows at the
 *                          matches -> merge_feat; call site matchers.py:50) as ONE product: y [nwin win^2, N] = window tokens W^T (+ bias) (+ window_bias
 *                          [nwin, N] broadcast over a window's tokens), the tokens read from the NHWC fine map feat [Bimg,Hf,Wf,C] in place: window w =
 *                          the win x win pixels around cell cell_ids[w] (centre ((cell / wc) stride, (cell % wc) stride)) of image img_ids[w], zero
 *                          outside the map (zero_row: C zeros the caller provides).  C = K a multiple of 64.  Replaces mfr_loftr_gather_windows + the
 *                          window tensor + mfr_gemm_* + a broadcast add. */
int mfr_gemm_f16x2_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids, const int32_t *cell_ids, int nwin, int wc, int stride, int win,
                           const float *zero_row, const void *packed_w, const float *bias, const float *window_bias, float *y, int ldy, int N, void *stream);
int mfr_gemm_bf16x3_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids, const int32_t *cell_ids, int nwin, int wc, int stride, int win,
                            const float *zero_row, const void *packed_w, const float *bias, const float *window_bias, float *y, int ldy, int N, void *stream);
/*   mfr_mlp_ln_f16x2 / mfr_mlp_ln_bf16x3 (round 6): y = [y +] LayerNorm_128( relu(x W1^T + b1) W2^T + b2 ) * gamma + beta in ONE launch: the MLP + norm2 +
 *                          residual of upstream LoFTREncoderLayer.forward at d_model 128 (`message = self.mlp(cat[x, message])`, `self.norm2`, `x + message`;
 *                          call site matchers.py:50).  x [M, K1] (K1 % 64 == 0; the fine level: 256 = [x | message]); W1 [256, K1] and W2 [128, 256] packed by
 *                          mfr_gemm_*_pack; the 256 hidden activations of a 128-row tile never leave the chip.  accumulate != 0: y += (in place residual). */
int mfr_mlp_ln_f16x2(const float *x, int ldx, int K1, const void *packed_w1, const float *b1, const void *packed_w2, const float *b2, const float *gamma, const float *beta,
                     float eps, float *y, int ldy, int M, int accumulate, void *stream);
int mfr_mlp_ln_bf16x3(const float *x, int ldx, int K1, const void *packed_w1, const float *b1, const void *packed_w2, const float *b2, const float *gamma, const float *beta,
                      float eps, float *y, int ldy, int M, int accumulate, void *stream);
size_t mfr_gemm_bf16x3_pack_bytes(int N, int K);
int mfr_gemm_bf16x3_pack(const float *w, int N, int K, void *packed, void *stream);
int mfr_gemm_bf16x3(const float *x, int ldx, const void *packed_w, const float *bias, float *y, int ldy, int M, int N, int K, int flags, void *stream);
int mfr_sg_attention(const float *q, const float *k, const float *v, int ld, int B2, int N, int heads,
                     const int32_t *n_tok, int cross, float *out, int ldo, void *stream);
int mfr_sg_attention_variant(const float *q, const float *k, const float *v, int ld, int B2, int N, int heads,
                             const int32_t *n_tok, int cross, float *out, int ldo, int variant, void *stream);
size_t mfr_sg_match_workspace_bytes(int B, int ldS);
int mfr_sg_sinkhorn_match(const float *S, int B, int ldS, const int32_t *n0, const int32_t *n1,
                          float bin_score, int iters, float match_thr,
                          const float *kpts0, const float *kpts1, int K,
                          void *workspace, size_t workspace_bytes,
                          int32_t *matches0, float *mscores0, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                          void *stream);
int mfr_sg_sinkhorn_match_variant(const float *S, int B, int ldS, const int32_t *n0, const int32_t *n1,
                                  float bin_score, int iters, float match_thr,
                                  const float *kpts0, const float *kpts1, int K,
                                  void *workspace, size_t workspace_bytes,
                                  int32_t *matches0, float *mscores0, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                                  int variant, void *stream);

/* ------------------------------------------------------------------------------------------
 * LoFTR kernels.  Reference call site: LoFTR_matcher.match, etc/feature_matching_baselines/
 * matchers.py:24-59 (LoFTR(default_cfg), *_ot.ckpt strict=False :16-18); network = un-vendored
 * zju3dv/LoFTR submodule (.gitmodules:1-3), restated per SURVEY.md Appendix A.4.
 *   mfr_loftr_linear_attention   LinearAttention of the coarse transformer (heads x 32, elu+1 feature
 *                                map, eps 1e-6): out = phi(Q) (phi(K)^T V/L) / (phi(Q).sum phi(K)) * L.
 *                                q [B,L,ldq], k,v [B,L,ld], out [B,L,ldo]; head h = channels [32h,32h+32)
 *   mfr_loftr_fine_attention     the same LinearAttention for the fine transformer: d_model 128 (8 heads x 16), L = 25
 *                                tokens per 5x5 window, Bw windows; one wavefront per window, no workspace.
 *                                q [Bw,25,ldq], k,v [Bw,25,ld], out [Bw,25,ldo]; L, D, heads must be 25, 128, 8
 *   mfr_loftr_coarse_match       CoarseMatching(dual_softmax) + get_coarse_match: S [B,L0,L1] =
 *                                (f0/sqrt C)(f1/sqrt C)^T; conf = softmax_i(S/T) * softmax_j(S/T);
 *                                conf > thr, border removal, mutual max -> i_ids, j_ids [B,L0] i32
 *                                (ascending i), mconf [B,L0], n_match [B]
 *   mfr_loftr_gather_windows     FinePreprocess unfold(win, stride, pad win/2) restricted to the
 *                                matched cells: feat [Bimg,Hf,Wf,C] NHWC -> out [M, win*win, C]
 *   mfr_loftr_fine_match         FineMatching: centre-feature correlation, softmax, spatial expectation,
 *                                sub-pixel update of the view-1 keypoints
 * ------------------------------------------------------------------------------------------ */
size_t mfr_loftr_linear_attention_workspace_bytes(int B, int L, int heads);
int mfr_loftr_linear_attention(const float *q, int ldq, const float *k, const float *v, int ld, int B, int L, int heads,
                               void *workspace, size_t workspace_bytes, float *out, int ldo, void *stream);
int mfr_loftr_fine_attention(const float *q, int ldq, const float *k, const float *v, int ld, int Bw, int L, int D, int heads,
                             float *out, int ldo, void *stream);
size_t mfr_loftr_coarse_match_workspace_bytes(int B, int L0, int L1);
int mfr_loftr_coarse_match(const float *S, int B, int h0, int w0, int h1, int w1, float temperature, float thr, int border,
                           void *workspace, size_t workspace_bytes, int32_t *i_ids, int32_t *j_ids, float *mconf,
                           int32_t *n_match, void *stream);
/* variant 0 (what mfr_loftr_coarse_match runs): S is swept twice (tiled kernels produce row AND column quantities in the same
 * sweep); variant 1: the four-sweep row / column kernels of round 1.  Same arithmetic per element; the partial logsumexps are
 * folded in a different (fixed) order. */
int mfr_loftr_coarse_match_variant(const float *S, int B, int h0, int w0, int h1, int w1, float temperature, float thr, int border,
                                   void *workspace, size_t workspace_bytes, int32_t *i_ids, int32_t *j_ids, float *mconf,
                                   int32_t *n_match, int variant, void *stream);
int mfr_loftr_gather_windows(const float *feat, int Bimg, int Hf, int Wf, int C, const int32_t *img_ids,
                             const int32_t *cell_ids, int M, int wc, int stride, int win, float *out, void *stream);
/* FineMatching (the last step of LoFTR_matcher.match, matchers.py:50-55 -> mkpts1_f): per matched window, similarity of view 0's centre
 * feature against the W*W fine features of view 1 (/ sqrt C), softmax, spatial expectation over linspace(-1, 1, W)^2, and
 * pts1[lin_idx[m]] = k1[lin_idx[m]] + expectation * out_scale (out_scale = (W / 2) * image-to-fine-map scale).
 * g0, g1 [M, W*W, ld] f32 (first C columns), lin_idx [M] i32 slots into the [B * L0, 2] tensors k1 / pts1; expec [M, 2] optional
 * (the normalised expectation; pts1 / lin_idx / k1 may be NULL when only expec is wanted).  W*W <= 64. */
int mfr_loftr_fine_match(const float *g0, const float *g1, int ld, int C, int M, int W, float out_scale, const int32_t *lin_idx,
                         const float *k1, float *pts1, float *expec, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused conv epilogues of the SuperPoint encoder (NCHW f32; same call site as above): the MIOpen
 * convolutions run without bias and these finish the layer in one pass.
 *   mfr_bias_relu_nchw        x <- relu(x + bias[c]) in place, x [B,C,HW]
 *   mfr_bias_pool2_relu_nchw  y = relu(max_pool2x2(x) + bias[c]), x [B,C,H,W] -> y [B,C,H/2,W/2]
 * Bit-identical to conv -> +bias -> ReLU -> max_pool2d(2,2) (monotone rounding).
 *   mfr_conv3x3_c1_relu       SuperPoint conv1a fused: y = relu(conv3x3(x [B,1,H,W], w [64,1,3,3], pad 1) + bias),
 *                             y [B,64,H,W]; W % 4 == 0 (HBM-write-bound first layer, one pass)
 */
int mfr_conv3x3_c1_relu(const float *x, const float *w, const float *bias, int B, int H, int W, int out_channels,
                        float *y, void *stream);
int mfr_bias_relu_nchw(float *x, const float *bias, int B, int C, int HW, void *stream);
/* 1x1 convolution with few output channels (SuperPoint's detector head convPb 256 -> 65, same call site): y [B,Cout,HW] = w [Cout,Cin] x [B,Cin,HW]
 * + bias, one ascending fused-multiply-add chain per output: a pixel's result does not depend on the batch size.  Cin <= 512. */
int mfr_conv1x1_nchw(const float *x, const float *w, const float *bias, int B, int Cin, int Cout, int HW, float *y, void *stream);
/*   mfr_nchw_to_rows       x [B, C, HW] (+ add [C, HW], may be NULL) -> out[img'][p][c] with row stride ldo and image stride out_img_stride
 *                          (floats): the token-major operand of the linear layers, written in place of torch's permute().contiguous();
 *                          deinterleave != 0: image 2 k + s of a pair-interleaved batch goes to slot s * (B / 2) + k. */
int mfr_nchw_to_rows(const float *x, const float *add, int B, int C, int HW, int deinterleave, float *out, long long out_img_stride, int ldo, void *stream);
int mfr_bias_pool2_relu_nchw(const float *x, const float *bias, int B, int C, int H, int W, float *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * ProcrustesSolver's optional whole-cloud ICP refinement (PROCRUSTES.REFINE, lib/models/matching/pose_solver.py:290-319;
 * config/matching/scannet/ *_icp.yaml): o3d registration_icp(point-to-point, max distance, init = RANSAC transform,
 * ICPConvergenceCriteria(rel_fitness, rel_rmse, max_iter)) restated (csrc/procrustes_icp.hip); R, t are in/out, pairs whose
 * `status` (may be NULL) is not MFR_ST_OK are left untouched with 0 inliers; n_inliers = int(fitness * |target cloud|) (:319).
 */
size_t mfr_procrustes_icp_workspace_bytes(int B, int H, int W);
int mfr_procrustes_icp_refine(const float *depth0, const float *depth1, int B, int H, int W, const void *K0, const void *K1, int k_dtype,
                              double max_corr_dist, double rel_fitness, double rel_rmse, int max_iter, const int32_t *status,
                              void *workspace, size_t workspace_bytes, double *R, double *t, int32_t *n_inliers, double *fitness,
                              double *rmse, int32_t *iters, void *stream);

/* ------------------------------------------------------------------------------------------
 * LoFTR transformer / FPN glue, fused (csrc/loftr_fused.hip).  Reference call site: LoFTR_matcher.match
 * (etc/feature_matching_baselines/matchers.py:24-59) -> upstream LoFTREncoderLayer / ResNetFPN_8_2 (un-vendored).
 *   mfr_layernorm       torch.nn.LayerNorm(C) over rows of x (row stride ldx floats), C in {128, 256}, + optional residual
 *                       (row stride ldr; may alias out): out[r] = residual[r] + (x[r] - mean) * rstd * gamma + beta, row stride
 *                       ldo.  Strides let norm1 write into the right half of the MLP's [x | message] operand (no cat) and
 *                       norm2 + residual update x in place.  All pointers 16-byte aligned, strides multiples of 4.
 *   mfr_upsample2x_add  y [planes,2H,2W] += F.interpolate(lo [planes,H,W], scale_factor=2, bilinear, align_corners=True)
 */
int mfr_layernorm(const float *x, int ldx, const float *gamma, const float *beta, const float *residual, int ldr, long long rows, int C,
                  float eps, float *out, int ldo, void *stream);
int mfr_upsample2x_add(const float *lo, float *y, int planes, int H, int W, void *stream);
/*   mfr_upsample_bilinear  out [planes,Ho,Wo] = F.interpolate(in [planes,H,W], size=(Ho,Wo), bilinear, align_corners=True) with
 *                       torch's arithmetic; dtype 0 = float32, 1 = bfloat16 storage (f32 arithmetic).  Reference call site: the
 *                       regression encoder's `upconv` (lib/models/regression/encoder/resunet.py:30-38, used at :121-126). */
int mfr_upsample_bilinear(const void *in, void *out, int planes, int H, int W, int Ho, int Wo, int dtype, void *stream);
/*   mfr_conv_gemm_bf16    the 3x3 decoder convolutions of the regression encoder (`conv` inside upconv4 / iconv4 / upconv3 / iconv3,
 *                       lib/models/regression/encoder/resunet.py:16-38, 112-128; bf16 autocast of train.py:20-70) as implicit GEMMs on
 *                       the bf16 matrix cores (csrc/conv_gemm_bf16.hip): forward, d/d input and d/d weight are ONE kernel,
 *                           C[i, j] = sum_k A[i, k] B[j, k]      bf16 operands, fp32 accumulate,
 *                       whose k axis is cut into segments of Lk elements (Lk % 32 == 0); chunk c of 32 elements, in segment s = 32 c / Lk,
 *                       is read at A + zA[z] + i * sA + segA[s] + (32 c mod Lk) and at B + zB[z] + j * sB + segB[s] + (32 c mod Lk)
 *                       (all offsets and strides in ELEMENTS, multiples of 8; segA / segB hold one entry MORE than there are segments).
 *                       Slice z of the grid works on chunks [zk[z], min(zk[z] + nkc_z, nkc_total)) and writes C + zC[z] (element
 *                       offset), row stride ldc; out_dtype 0 = float32, 1 = bfloat16 (round to nearest even); bias [N] f32 (added
 *                       before rounding) or NULL; zA / zB / zC / zk may be NULL (= 0).  Rows i >= M / j >= N of a tile are neither read
 *                       nor written.  How a convolution maps onto it: map-free-reloc_amd/regression/conv_bf16.py. */
int mfr_conv_gemm_bf16(const void *A, long long sA, const long long *segA, const void *B, long long sB, const long long *segB,
                       int Lk, int nkc_total, int nkc_z, const float *bias, void *C, long long ldc, int out_dtype,
                       int M, int N, int nz, const long long *zA, const long long *zB, const long long *zC, const int *zk, void *stream);
/*   operand images of those products (memory-bound, every element written incl. the zero halo; x_dtype 0 = float32, 1 = bfloat16):
 *   mfr_conv_pack_nhwc_halo  x [B,C,H,W] -> out = guard_rows * C zeros | [B, H+2, Wp, C] bf16 | guard_rows * C zeros   (C % 8 == 0); Wp = W+2: a
 *                            zero column on either side of a row; Wp = W+1: one zero column in front of every row, shared with the row before
 *   mfr_conv_pack_cm_halo    x [BC,H,W] -> ncopies images, each  slack zeros | [BC][L] bf16 | slack zeros,  rows of Wq >= W+2 positions
 *                            (Wq % 8 == 0, L >= (H+2) Wq, L % 8 == 0, slack % 8 == 0); copy k holds the haloed image shifted by
 *                            first_shift + k positions along its rows (the kx tap shift of the d/d weight product) */
int mfr_conv_pack_nhwc_halo(const void *x, int x_dtype, int B, int C, int H, int W, int Wp, void *out, int guard_rows, void *stream);
int mfr_conv_pack_cm_halo(const void *x, int x_dtype, long long BC, int H, int W, int Wq, long long L, int ncopies, int first_shift, long long slack,
                          void *out, void *stream);
/*   mfr_conv_unpack_nchw    haloed NHWC result [B, H+2, Wp, N] bf16 (what the forward / d input product writes) -> y [B, N, H, W] bf16 */
int mfr_conv_unpack_nchw(const void *haloed, int B, int N, int H, int W, int Wp, void *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * Differentiable batched Kabsch rotation of the regression heads (csrc/kabsch.hip, csrc/kabsch_math.h).  Reference: `procrustes`,
 * lib/utils/solver.py:4-37 (torch.svd + reflection fix), called from lib/models/regression/head.py:55-163.  No host
 * synchronisation (torch.linalg.svd reads the solver's status back), so the training step can be captured as a HIP graph.
 *   mfr_kabsch_fwd   H [B,3,3] = A_c^T B_c -> R [B,3,3], the proper rotation with B_c ~ A_c R^T (Horn quaternion + Jacobi)
 *   mfr_kabsch_bwd   gR = dL/dR -> gH = dL/dH (closed form through the polar factor, see kabsch.hip)
 * ------------------------------------------------------------------------------------------ */
int mfr_kabsch_fwd(const float *H, int B, float *R, void *stream);
int mfr_kabsch_bwd(const float *H, const float *gR, int B, float *gH, void *stream);

/* ------------------------------------------------------------------------------------------
 * SIFT-descriptor correspondence leg.  Reference call sites:
 * SIFTMatching.get_correspondences (lib/models/matching/feature_matching.py:75-118) and
 * SIFT_matcher.match (etc/feature_matching_baselines/matchers.py:135-188), everything AFTER
 * cv.SIFT.detectAndCompute (keypoint detection/description stays with the caller):
 *   mfr_rootsift          root_sift (feature_matching.py:68-74): out = sqrt(desc / (rowsum + 1e-7f)),
 *                         desc/out [n_rows,128] f32 (out may alias desc), norm2 [n_rows] = |out row|^2.
 *                         Bit-identical to the reference's numpy (same summation order).
 *   mfr_desc_ratio_match  replaces cv.FlannBasedMatcher(kd-tree).knnMatch(des0, des1, k=2) + Lowe's
 *                         ratio loop (:86-101): EXACT 2-NN (squared L2, ties -> lower train index) of
 *                         every query row, then keep i where sqrt(d1) < ratio * sqrt(d2) (binary64
 *                         compare, as the Python loop does), in query order.
 *                         des0 [B,N0,128], des1 [B,N1,128] (rootSIFT), norm0 [B,N0], norm1 [B,N1],
 *                         kp0 [B,N0,2], kp1 [B,N1,2] pixel (x,y); n0,n1 [B] valid rows per pair.
 *                         Out: nn_idx [B,N0] i32, nn_d2 [B,N0,2] f32 (best, second), pts0/pts1
 *                         [B,maxN,2] + n_corr [B] in the layout the solver entry points take.
 *                         Pairs with n1 < 2 yield no correspondences.
 * ------------------------------------------------------------------------------------------ */
int mfr_rootsift(const float *desc, int n_rows, float *out, float *norm2, void *stream);
int mfr_desc_ratio_match(const float *des0, const float *des1, const float *norm0, const float *norm1,
                         const float *kp0, const float *kp1, int B, int N0, int N1,
                         const int32_t *n0, const int32_t *n1, double ratio,
                         int32_t *nn_idx, float *nn_d2, float *pts0, float *pts1, int maxN, int32_t *n_corr,
                         void *stream);

/* ------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 convolutions of the SuperPoint encoder (conv1b..conv4b, convPa, convDa; same
 * call site as the SuperPoint kernels above) as a fused Winograd F(2x2,3x3) kernel on the fp32 matrix
 * cores, with +bias, ReLU and the optional 2x2 max-pool in the output transform (csrc/winograd_conv.hip).
 *   mfr_wino_filter_bytes      size of the packed transformed filter (16*Cin*ceil32(Cout) floats); 0 if unsupported
 *   mfr_wino_filter_transform  w [Cout,Cin,3,3] f32 -> upk (G g G^T, packed in MFMA operand order); once per weight set
 *   mfr_conv3x3_wino           x [B,Cin,H,W] -> y [B,Cout,H,W] (pool=0) or [B,Cout,H/2,W/2] (pool=1; = max_pool2d(2,2)
 *                              of the activation); y = act(conv(x) + bias + residual): bias, residual ([B,Cout,H,W],
 *                              pool=0 only) may be NULL; act 0 none, 1 ReLU, 2 LeakyReLU(0.01).  Cin % 4 == 0; any Cout
 *                              (padded to a multiple of 32 inside the packed filter).  Also serves the stride-1 3x3
 *                              convolutions of LoFTR's ResNet-FPN backbone (BatchNorm folded into w / bias).
 * f32 Winograd arithmetic: agrees with a direct f32 convolution to ~1e-6 relative (not bit-identical).
 * ------------------------------------------------------------------------------------------ */
/* The same layer (same arguments, same epilogue) on the 16-bit matrix cores at fp32 accuracy by operand splitting (csrc/winograd_split.hip):
 * U = G g G^T and V = B^T d B are formed in fp32 as above.  f16x2 (round 5, what nets/conv.py runs; csrc/split_f16.h): V as two f16 terms,
 * U pre-scaled per output channel and packed as three f16 terms, three partial products; precondition |activation| < 16376.
 * bf16x3 (rounds 3-4): every operand split exactly into three bf16 terms, six partial products.  fp32 accumulate either way; error vs fp64 =
 * that of the exact-fp32 matrix instruction (profiles/r05_f16x2_probe.jsonl, r03_bf16x3_probe.jsonl).  Any Cin, Cout (padded to multiples
 * of 16 / 64 inside the packed filter; the two packed formats are not interchangeable).
 *   mfr_wino_*_filter_bytes      size of the packed, split filter (ceil(Cout/64) * ceil(Cin/16) * 96 KiB; f16x2: + the channel scales)
 *   mfr_wino_*_filter_transform  w [Cout,Cin,3,3] f32 -> upk; once per weight set
 *   mfr_conv3x3_wino_*           as mfr_conv3x3_wino */
size_t mfr_wino_f16x2_filter_bytes(int Cin, int Cout);
int mfr_wino_f16x2_filter_transform(const float *w, int Cin, int Cout, void *upk, void *stream);
int mfr_conv3x3_wino_f16x2(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout,
                           int H, int W, int act, int pool, float *y, void *stream);
/* The same layer once more (same arguments, same epilogue, f16x2 arithmetic) as a DIRECT implicit GEMM (round 6, csrc/conv_direct.hip): the halo tile
 * of 16 input channels is split ONCE into LDS in operand form and the nine taps are shifted reads of it; a 4-wavefront workgroup (two per CU) computes 16 x 32 pixels x 64
 * channels (Cout <= 64) or 8 x 32 pixels x 128 channels.  2.25x Winograd's matrix instructions at ~2 other instructions per MFMA instead of 10-18: 1.0-1.3x
 * the Winograd kernel on every layer of the two backbones (profiles/r06_ab_direct_conv_halo.json), the default of nets/conv.py for SPLIT = f16x2.  Precondition |activation| <= 65504 (guarded: mfr_f16x2_guard_bind).
 *   mfr_conv3x3_direct_f16x2_filter_bytes   size of the packed filter: ceil(Cout/64) * ceil(Cin/16) * 54 KiB + the channel scales
 *   mfr_conv3x3_direct_f16x2_filter_pack    w [Cout,Cin,3,3] f32 -> packed; once per weight set
 *   mfr_conv3x3_direct_f16x2                as mfr_conv3x3_wino */
size_t mfr_conv3x3_direct_f16x2_filter_bytes(int Cin, int Cout);
int mfr_conv3x3_direct_f16x2_filter_pack(const float *w, int Cin, int Cout, void *packed, void *stream);
int mfr_conv3x3_direct_f16x2(const float *x, const void *packed, const float *bias, const float *residual, int B, int Cin, int Cout,
                             int H, int W, int act, int pool, float *y, void *stream);
/*   mfr_conv3x3_direct_f16x2_rows           the same layer with TOKEN-MAJOR output: yrows [B H W, ldy], channel c of pixel p at yrows[p ldy + c] (ldy >= Cout, both multiples
 *                                           of 4) -- what the 1x1 / linear layer behind the convolution reads (SuperPoint convDa -> convDb, models/superpoint.py; LoFTR
 *                                           layer1_outconv2 -> FinePreprocess' unfold, matchers.py:50): replaces the NCHW store + mfr_nchw_to_rows.  No residual, no pool. */
int mfr_conv3x3_direct_f16x2_rows(const float *x, const void *packed, const float *bias, int B, int Cin, int Cout, int H, int W, int act, float *yrows, int ldy, void *stream);
/*   mfr_conv3x3s2_direct_f16x2              the STRIDE-2 3x3 / pad 1 layer (LoFTR's layer2.0 / layer3.0 conv1, `nn.Conv2d(k=3, s=2, p=1)` + folded BatchNorm + ReLU;
 *                                           un-vendored loftr/backbone/resnet_fpn.py, call site matchers.py:50) through the same kernel: the (2 TR + 1) x 65 patch is staged
 *                                           with its columns de-interleaved by parity, so every tap is again 32 consecutive LDS units; same packed filter as the
 *                                           stride-1 entry.  y [B,Cout,(H-1)/2+1,(W-1)/2+1]; act as above; no residual, no pool.  Rounds 1-5: mfr_conv_igemm_f16x2. */
int mfr_conv3x3s2_direct_f16x2(const float *x, const void *packed, const float *bias, int B, int Cin, int Cout, int H, int W, int act, float *y, void *stream);
/*   mfr_sp_conv1ab_f16x2         SuperPoint's first two layers in ONE kernel (round 5): y [B,64,H/2,W/2] = max_pool2(relu(conv1b(relu(conv1a(gray))))) for
 *                                gray [B,1,H,W]; w1a [64,1,3,3] / b1a [64] as they are, upk1b = mfr_wino_f16x2_filter_transform of conv1b's [64,64,3,3].
 *                                Bit-identical to mfr_conv3x3_c1_relu followed by mfr_conv3x3_wino_f16x2(..., act 1, pool 1): the 64-channel
 *                                intermediate (6.4 GB per 64 images) never exists in memory. */
int mfr_sp_conv1ab_f16x2(const float *gray, const float *w1a, const float *b1a, const void *upk1b, const float *bias1b, int B, int H, int W, float *y, void *stream);
size_t mfr_wino_bf16x3_filter_bytes(int Cin, int Cout);
int mfr_wino_bf16x3_filter_transform(const float *w, int Cin, int Cout, void *upk, void *stream);
int mfr_conv3x3_wino_bf16x3(const float *x, const void *upk, const float *bias, const float *residual, int B, int Cin, int Cout,
                            int H, int W, int act, int pool, float *y, void *stream);
size_t mfr_wino_filter_bytes(int Cin, int Cout);
int mfr_wino_filter_transform(const float *w, int Cin, int Cout, float *upk, void *stream);
int mfr_conv3x3_wino(const float *x, const float *upk, const float *bias, const float *residual, int B, int Cin, int Cout,
                     int H, int W, int act, int pool, float *y, void *stream);
/* the same convolution through a named kernel variant: 0 default, 1 classic, 2 software-pipelined K loop (bit-identical to 1),
 * 4 shared-transform (two cout slices split every patch transform; bias folded into an accumulator: f32-roundoff differences);
 * other values are timing ablations of tools/tune_wino.py -- for A/B timing and parity tests */
int mfr_conv3x3_wino_variant(const float *x, const float *upk, const float *bias, const float *residual, int B, int Cin, int Cout,
                             int H, int W, int act, int pool, int variant, float *y, void *stream);

/* ------------------------------------------------------------------------------------------
 * Relative-pose-regression aggregator (csrc/corr_warp.hip).  Reference call sites:
 * CorrelationVolumeWarping.forward (lib/models/regression/aggregator.py:42-116) and
 * CorrelationVolumeWarpingQKV.forward (aggregator.py:134-191), and autograd's backward through them in
 * RegressionModel.training_step (lib/models/regression/model.py:87-97):
 *     cvolume = softmax(q^T k, dim=2);  warped = v cvolume^T;  pos = grid cvolume^T;  max = max_j cvolume
 * computed flash-style: the [B, N, N] volume never exists in memory, forward or backward.
 *   mfr_corr_warp_fwd   q, k [B,Dq,N] (Dq 16 or 32), v [B,32,N], grid [2,N] or NULL -> warped [B,32,N], pos [B,2,N]
 *                       (NULL iff grid NULL), max_score [B,N]; row_max / row_sum [B,N] are the softmax statistics
 *                       (max of log2(e)*score, and the sum of exp) the backward needs.
 *   mfr_corr_warp_bwd   d_warped [B,32,N], d_pos [B,2,N] or NULL, d_max [B,N] or NULL, delta [B,N] =
 *                       sum_c d_warped*warped + sum d_pos*pos + d_max*max_score  ->  dq, dk [B,Dq,N], dv [B,32,N].
 *                       One owner per output element (two kernels), no atomics: deterministic.
 * fp32 on the exact-fp32 matrix cores; agrees with the materialised fp32 computation to f32 round-off.
 * ------------------------------------------------------------------------------------------ */
int mfr_corr_warp_fwd(const float *q, const float *k, const float *v, const float *grid, int B, int Dq, int N,
                      float *warped, float *pos, float *max_score, float *row_max, float *row_sum, void *stream);
int mfr_corr_warp_bwd(const float *q, const float *k, const float *v, const float *grid, int B, int Dq, int N,
                      const float *d_warped, const float *d_pos, const float *d_max, const float *delta,
                      const float *row_max, const float *row_sum, float *dq, float *dk, float *dv, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MFR_HIP_H */
