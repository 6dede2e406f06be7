/*
 * oracle/mfr_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the pose-solver leg of nianticlabs/map-free-reloc
 * (reference: lib/models/matching/pose_solver.py).  Nothing under oracle/ is
 * ever imported, linked or executed by the product package
 * (map-free-reloc_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - depth gather / back-projection / scale-RANSAC / validity rules: pinned
 *     against the reference's own Python executed in the build container
 *     (oracle/gen_golden.py -> tests/golden/ ref_*.npz files).
 *   - PnP-RANSAC, E-matrix RANSAC, recoverPose: the arithmetic lives in
 *     opencv-python==4.8.0.74 (environment.yml:17), which is NOT available
 *     offline -> restated from the published algorithms, PARITY UNPINNED
 *     against OpenCV itself.  Pinned only by known-answer synthetic geometry.
 *
 * Floating-point contract shared with the HIP kernels (for bit-exact inlier
 * sets): IEEE binary64, only + - * / sqrt and comparisons, no FMA contraction
 * (-ffp-contract=off), reductions in "wave64 order" (64 strided partial sums,
 * then an xor-butterfly 32,16,8,4,2,1).
 */
#ifndef MFR_ORACLE_H
#define MFR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-pair status codes (same values as include/mfr_hip.h) */
enum {
    MFR_ST_OK = 0,
    MFR_ST_TOO_FEW = 1,      /* fewer correspondences than the solver minimum */
    MFR_ST_BAD_DEPTH = 2,    /* too few correspondences with valid depth      */
    MFR_ST_NO_MODEL = 3,     /* RANSAC found no model / refinement failed     */
    MFR_ST_DEGENERATE = 4    /* |t| > 1000 (pose_solver.py:223-225)           */
};

/* Intrinsics cross every boundary as 9 row-major values of the dtype the `data` dict holds (same tags as include/mfr_hip.h):
 *   MFR_K_F64  the Map-free loader's flow: correct_intrinsic_scale multiplies a float64 np.eye(3) into K
 *              (lib/datasets/utils.py:117-130, always called: lib/datasets/mapfree.py:50-52), so data['K_color*'] is
 *              float64 and np.linalg.inv(K) (pose_solver.py:16), the K-normalisation (:39-40) and the threshold mean
 *              (:43) are float64 arithmetic;
 *   MFR_K_F32  resize=None datasets: K stays float32, the inverse / normalisation / mean are float32 arithmetic
 *              (quirk Q5) and only the products are promoted to float64. */
#define MFR_K_F32 0
#define MFR_K_F64 1
typedef struct {
    double fx, fy, cx, cy;        /* as stored (f32 values widen exactly)                                  */
    double ifx, icx, ify, icy;    /* np.linalg.inv(K)[0,0], [0,2], [1,1], [1,2] evaluated in K's own dtype  */
    int f32;
} mfr_intr;
/* -1 if K is not a zero-skew pinhole matrix with bottom row [0,0,1] */
int mfr_ref_load_intr(const void *K, int k_dtype, mfr_intr *out);

void mfr_ref_philox4x32_10(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4]);
void mfr_ref_sample_distinct(uint64_t seed, uint64_t pair_id, uint32_t iter, int n, int k, int *out);
double mfr_ref_det_log(double x);
int mfr_ref_update_num_iters(double p, double ep, int model_points, int max_iters);
int mfr_ref_poly_real_roots(const double *c, int deg, double *roots);

/* pose_solver.py:6-17 -- xyz[N,3] = depth * (inv(K) @ [u,v,1]); K is the 3x3 pinhole matrix (zero skew) in
 * float32 or float64; the inverse is evaluated in K's dtype exactly as np.linalg.inv does (see mfr_ref_load_intr). */
int mfr_ref_backproject(const int32_t *uv, const float *depth, int n, const void *K, int k_dtype, double *xyz);

float mfr_ref_depth_min(const float *depth, int hw);

/* pose_solver.py:186-206 (PnP input prep): int-truncate pts0, gather depth0,
 * valid = d > depth_min, compact, back-project with K0.  obs = pts1 (f64). */
int mfr_ref_pnp_lift(const float *pts0, const float *pts1, int n, const float *depth0, int H, int W,
                     const void *K0, int k_dtype, double *xyz, double *obs, int32_t *src_idx);

/* cv.solvePnPRansac(P3P) + refit + cv.solvePnPGeneric(ITERATIVE) restatement
 * (pose_solver.py:209-235).  Returns per-pair status. */
int mfr_ref_pnp_ransac(const double *xyz, const double *obs, int n, const void *K1, int k_dtype,
                       int max_iters, double thr, double conf, uint64_t seed, uint64_t pair_id,
                       double R[9], double t[3], uint8_t *mask, int *n_inl,
                       int *best_iter, int *iters_run, int32_t *counts /* [max_iters] or NULL */);

/* full PnPSolver.estimate_pose (pose_solver.py:184-235) */
int mfr_ref_pnp_solve(const float *pts0, const float *pts1, int n, const float *depth0, int H, int W,
                      const void *K0, const void *K1, int k_dtype, int max_iters, double thr, double conf,
                      uint64_t seed, uint64_t pair_id, double R[9], double t[3], int *n_inl);

int mfr_ref_p3p(const double X[9], const double f[9], double Rs[36], double ts[12]);

/* pose_solver.py:137-172 -- depth lift of E-mat inliers + exhaustive 1-D scale RANSAC */
int mfr_ref_scale_lift(const float *pts0, const float *pts1, const uint8_t *mask, int n,
                       const float *depth0, const float *depth1, int H, int W,
                       const void *K0, const void *K1, int k_dtype, const double R[9], const double t[3],
                       double *scale /* [n] */);
int mfr_ref_scale_ransac(const double *scale, int n, double thr, double *best_scale, int *best_idx);

/* E-matrix model quality (same values as include/mfr_hip.h) */
#define MFR_EMAT_SCORE_MAGSAC 0   /* MAGSAC++ loss + sigma-consensus++ (what cv.USAC_MAGSAC names, pose_solver.py:46-48) */
#define MFR_EMAT_SCORE_COUNT  1   /* inlier count + LM polish (rounds 1-3) */
#define MFR_MAGSAC_LUT_M 2048     /* intervals of the loss / weight table the product ships with */
void mfr_ref_magsac_lut(double *lut /* [2 * (M + 1)] */, int M);
void mfr_ref_orthonormalize(double R[9]);

/* LM refinement used by the PnP path (exposed for tests) */
int mfr_ref_pnp_lm(const double *xyz, const double *obs, const int32_t *idx, int n_idx,
                   const double Kd[4], int max_iter, double R[9], double t[3]);

#ifdef __cplusplus
}
#endif
#endif
