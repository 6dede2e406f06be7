/*
 * oracle/mfr_oracle_desc.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; never imported by the product).
 *
 * rootSIFT + 2-nearest-neighbour + Lowe ratio test of the reference's SIFT correspondence path:
 *   SIFTMatching.get_correspondences  lib/models/matching/feature_matching.py:75-118
 *   SIFT_matcher.match                etc/feature_matching_baselines/matchers.py:135-188
 *
 * Parity status:
 *   - root_sift (feature_matching.py:68-74): PINNED bit-exactly against the reference's own numpy code
 *     executed in the build container (oracle/gen_golden.py -> tests/golden/ref_rootsift.npz).
 *   - ratio-test loop (:97-101): PINNED by the same fixture (the loop is replayed by the generator on
 *     a brute-force 2-NN table).
 *   - the 2-NN search itself is cv.FlannBasedMatcher (kd-tree forest, 5 trees, 50 checks;
 *     opencv-python==4.8.0.74, environment.yml:17 -- not available offline): an APPROXIMATE search whose
 *     result depends on its internal RNG.  Restated as the EXACT 2-NN it approximates; PARITY UNPINNED
 *     against FLANN itself (FLANN can miss a true neighbour; it cannot return a closer one).
 *
 * Arithmetic: f32 as in OpenCV/numpy.  d^2 = (|a|^2 + |b|^2) - 2 a.b with a.b an fmaf chain over
 * d = 0..127; the HIP kernel contracts in a different order on the matrix cores, so distances agree to
 * a few ulp and indices agree except on near-ties (tests compare with that margin).
 */
#include <math.h>
#include <stdint.h>

#define D 128

/* numpy pairwise_sum for a contiguous 128-element f32 row: 8 strided accumulators + 3-level tree
 * (numpy/_core/src/umath/loops_utils.h.src, n <= PW_BLOCKSIZE branch) */
static float row_sum_numpy(const float *a)
{
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (int i = 8; i < D; i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

/* feature_matching.py:68-74: descs /= (descs.sum(axis=1, keepdims=True) + 1e-7); descs = sqrt(descs) */
void mfr_ref_rootsift(const float *in, int n_rows, float *out, float *norm2)
{
    for (int r = 0; r < n_rows; ++r) {
        const float *a = in + (long)r * D;
        float *o = out + (long)r * D;
        const float den = row_sum_numpy(a) + 1e-7f;
        float n2 = 0.f;
        for (int d = 0; d < D; ++d) {
            o[d] = sqrtf(a[d] / den);
            n2 = fmaf(o[d], o[d], n2);
        }
        if (norm2) norm2[r] = n2;
    }
}

/* exact 2-NN (squared L2) of each of the n0 query rows among the n1 train rows; ties -> lower index.
 * nn_idx [n0], nn_d2 [n0,2] (best, second best; +inf when fewer than 1 / 2 train rows) */
void mfr_ref_desc_2nn(const float *des0, const float *des1, const float *nrm0, const float *nrm1,
                      int n0, int n1, int32_t *nn_idx, float *nn_d2)
{
    for (int i = 0; i < n0; ++i) {
        const float *a = des0 + (long)i * D;
        float b1 = INFINITY, b2 = INFINITY;
        int i1 = -1;
        for (int j = 0; j < n1; ++j) {
            const float *b = des1 + (long)j * D;
            float dot = 0.f;
            for (int d = 0; d < D; ++d) dot = fmaf(a[d], b[d], dot);
            float d2 = (nrm0[i] + nrm1[j]) - 2.f * dot;
            if (d2 < 0.f) d2 = 0.f;
            if (d2 < b1) { b2 = b1; b1 = d2; i1 = j; }
            else if (d2 < b2) b2 = d2;
        }
        nn_idx[i] = i1; nn_d2[2 * i] = b1; nn_d2[2 * i + 1] = b2;
    }
}

/* feature_matching.py:97-101: for (m, n) in matches: if m.distance < ratio * n.distance: keep.
 * DMatch.distance is the f32 L2 distance (sqrt of FLANN's squared L2); the Python comparison runs in
 * binary64.  kp0 [n0,2], kp1 [n1,2]; pts0/pts1 [>=n0, 2].  Returns the number of kept matches. */
int mfr_ref_desc_ratio(const int32_t *nn_idx, const float *nn_d2, int n0, int n1, double ratio,
                       const float *kp0, const float *kp1, float *pts0, float *pts1)
{
    int m = 0;
    if (n1 < 2) return 0;
    for (int i = 0; i < n0; ++i) {
        const float d1 = sqrtf(nn_d2[2 * i]), d2 = sqrtf(nn_d2[2 * i + 1]);
        if ((double)d1 < ratio * (double)d2) {
            const int j = nn_idx[i];
            pts0[2 * m] = kp0[2 * i]; pts0[2 * m + 1] = kp0[2 * i + 1];
            pts1[2 * m] = kp1[2 * j]; pts1[2 * m + 1] = kp1[2 * j + 1];
            ++m;
        }
    }
    return m;
}
