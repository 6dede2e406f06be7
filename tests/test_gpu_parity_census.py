"""-m gpu: end-to-end parity census at POSE level (north-star bar: bit-exact inlier indices for a fixed RANSAC seed, pose
within 1e-4 rad / 1e-4 m on the same pairs).  The whole HIP pipeline (matcher -> lift -> RANSAC) against the whole
CPU-oracle pipeline (oracle/pipeline_ref.py) on 32 synthetic pairs for configs[1] and 8 for configs[2]; nothing is
masked: every pair is counted, the identical-match-set fraction is printed and bounded from below."""
import json

import pytest

from tools.parity_census import census
from oracle import pipeline_ref as PR

pytestmark = pytest.mark.gpu


def _check(recs, min_identical, min_common):
    s = PR.summarize(recs)
    print(json.dumps(s))
    # every pair solved by one side is solved by the other
    assert s["status_agree"] == s["pairs"], [r for r in recs if r["status_ref"] != r["status_hip"]]
    # wherever the HIP matcher reproduces the oracle's match set exactly, the solver stage is bit-exact: same pose, same count
    for r in recs:
        if r["identical_matches"] and "rot_rad" in r:
            assert r["rot_rad"] <= 1e-4 and r["trans_m"] <= 1e-4 and r["inliers_ref"] == r["inliers_hip"], r
    assert s["identical_fraction"] >= min_identical, s
    assert s["mean_common_fraction"] >= min_common, s
    return s


def test_census_superglue_pnp_32_pairs():
    s = _check(census("sg_pnp", [5000 + i for i in range(32)]), SG_MIN_IDENTICAL, 0.97)
    assert s["identical_as_sets"] >= 28 and s["inlier_count_equal"] >= 28 and s["pose_within_bar"] >= 30, s     # measured: 32 / 32 / 32
    # fp32 near-tie decisions (keypoint order at equal scores, Sinkhorn scores next to the 0.2 threshold) may flip a match in
    # some pairs; the pose must still be the oracle's far below the benchmark's resolution (0.25 m / 5 deg)
    assert s["max_rot_rad"] < 5e-3 and s["max_trans_m"] < 5e-3, s


def test_census_loftr_emat_8_pairs():
    s = _check(census("loftr_emat", [5000 + i for i in range(8)], chunk=4), LOFTR_MIN_IDENTICAL, 0.99)
    assert s["pose_within_bar"] >= 7 and s["inlier_count_equal"] >= 6, s                                        # measured: 8 / 8
    assert s["max_rot_rad"] < 2e-2 and s["max_trans_m"] < 2e-2, s


def test_census_hard_scenes_inlier_index_sets():
    """hard scenes (moving objects + occluder, and for PnP 35-55 % of the depth map wrong): 30-60 % outliers, the RANSACs run hundreds of hypotheses.
    Wherever the matcher reproduces the oracle's match set, the inlier INDEX SET (canonical order) is the oracle's, bit for bit."""
    recs = census("sg_pnp", [5000 + i for i in range(16)], hard=2)
    s = PR.summarize(recs)
    print(json.dumps(s))
    assert s["status_agree"] == s["pairs"]
    assert s["inlier_index_sets_compared"] >= 14 and s["median_inlier_fraction"] < 0.8, s           # 35-55 % of the lifted points carry a wrong depth
    for r in recs:
        if r["identical_set"] and "inlier_set_identical" in r:
            assert r["inlier_set_identical"] and r["rot_rad"] <= 1e-4 and r["trans_m"] <= 1e-4, r
    assert s["pose_within_bar"] >= 13, s
    recs = census("loftr_emat", [5000 + i for i in range(4)], chunk=4, hard=True)
    s = PR.summarize(recs)
    print(json.dumps(s))
    assert s["status_agree"] == s["pairs"] and s["median_inlier_fraction"] < 0.8, s                    # 30-60 % outliers for the E-mat RANSAC
    assert s["min_inlier_set_jaccard_q64"] > 0.97 and s["pose_within_bar"] == s["pairs"], s


def test_census_procrustes_and_sift_leg():
    """f-1 (SuperGlue -> Procrustes RANSAC) and configs[0] (descriptor leg -> E-mat RANSAC): whole HIP path vs whole oracle path"""
    s = PR.summarize(census("sg_procrustes", [5000 + i for i in range(8)], hard=2))
    print(json.dumps(s))
    assert s["status_agree"] == s["pairs"] and s["pose_within_bar"] >= 6 and s["inlier_count_equal"] >= 6, s
    recs = census("sift_emat", list(range(8)))
    s = PR.summarize(recs)
    print(json.dumps(s))
    assert s["identical_match_sets"] == 8 and s["inlier_index_sets_identical"] == 8 and s["pose_bit_equal"] == 8, s


# lower bounds on the fraction of pairs whose whole match set is bit-identical to the oracle's (measured:
# profiles/r02_parity_census.json); LoFTR's fine stage is a sub-pixel fp32 expectation, so exact equality of every
# coordinate is not expected there and the common-fraction (1/64 px quantised) carries the check
SG_MIN_IDENTICAL = 0.4        # measured 0.66: the other pairs hold the same SET in another keypoint order (equal-score ties)
LOFTR_MIN_IDENTICAL = 0.0
