"""-m gpu parity: rootSIFT + exact 2-NN + ratio test (csrc/descriptor_match.hip) vs the CPU oracle
(oracle/mfr_oracle_desc.c) and vs the fixture produced by the reference's own get_correspondences loop
(tests/golden/ref_sift_ratio.npz).  rootSIFT: bit-exact.  2-NN: the matrix cores contract the 128-d dot
product in a different order than the oracle's fmaf chain, so squared distances agree to 1e-6 and indices
must agree wherever the oracle's decision margin exceeds that."""
import os

import numpy as np
import pytest
import torch

from mapfree_reloc_amd import descriptor_ops as D
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-6


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _sift_like(rng, n):
    d = rng.gamma(0.6, 1.0, (n, 128))
    d = np.minimum(d / np.linalg.norm(d, axis=1, keepdims=True), 0.2)
    return np.clip(np.rint(512.0 * d / np.linalg.norm(d, axis=1, keepdims=True)), 0, 255).astype(np.float32)


def test_rootsift_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_sift_ratio.npz"))
    for name in ("rs_int", "rs_float"):
        out, n2 = D.rootsift(_dev(g[name + "_in"]))
        np.testing.assert_array_equal(out.cpu().numpy(), g[name + "_out"])
        _, n2_ref = O.rootsift(g[name + "_in"])
        np.testing.assert_array_equal(n2.cpu().numpy(), n2_ref)


def test_reference_loop_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_sift_ratio.npz"))
    m = D.DescriptorRatioMatcher(0.8, DEV)
    f0 = [(g[f"gc{c}_kp0"], g[f"gc{c}_des0"].astype(np.float32)) for c in range(3)]
    f1 = [(g[f"gc{c}_kp1"], g[f"gc{c}_des1"].astype(np.float32)) for c in range(3)]
    out = m(f0, f1)                                     # ragged batch: 300/64/2048 x 280/97/2048
    for c in range(3):
        n = int(out["n_corr"][c])
        np.testing.assert_array_equal(out["pts0"][c, :n].cpu().numpy(), g[f"gc{c}_pts1"])
        np.testing.assert_array_equal(out["pts1"][c, :n].cpu().numpy(), g[f"gc{c}_pts2"])


@pytest.mark.parametrize("n0,n1", [(1, 2), (33, 31), (128, 129), (700, 1500), (2048, 2048)])
def test_2nn_vs_oracle(n0, n1):
    rng = np.random.default_rng(n0 * 7 + n1)
    d0, d1 = _sift_like(rng, n0), _sift_like(rng, n1)
    k = min(n0, n1) // 2
    d1[:k] = np.clip(np.rint(d0[:k] + rng.normal(0, rng.uniform(2, 60, (k, 1)), (k, 128))), 0, 255)
    kp0 = (rng.random((n0, 2)) * 700).astype(np.float32); kp1 = (rng.random((n1, 2)) * 700).astype(np.float32)
    r0, q0 = O.rootsift(d0); r1, q1 = O.rootsift(d1)
    idx_ref, d2_ref = O.desc_2nn(r0, r1, q0, q1)
    out = D.DescriptorRatioMatcher(0.8, DEV)([(kp0, d0)], [(kp1, d1)])
    idx = out["nn_idx"][0, :n0].cpu().numpy(); d2 = out["nn_d2"][0, :n0].cpu().numpy()
    np.testing.assert_allclose(d2, d2_ref, rtol=0, atol=TOL)
    clear = (d2_ref[:, 1] - d2_ref[:, 0]) > 2 * TOL
    np.testing.assert_array_equal(idx[clear], idx_ref[clear])
    # ratio decision and ordered compaction: replay the oracle's loop on the GPU's own 2-NN table (exact), and
    # require the same kept set as the oracle wherever the oracle's margin is clear
    p0, p1 = O.desc_ratio(idx, d2, n1, 0.8, kp0, kp1)
    n = int(out["n_corr"][0])
    np.testing.assert_array_equal(out["pts0"][0, :n].cpu().numpy(), p0)
    np.testing.assert_array_equal(out["pts1"][0, :n].cpu().numpy(), p1)
    s = np.sqrt(d2_ref.astype(np.float64))
    margin = np.abs(s[:, 0] - 0.8 * s[:, 1]) > 1e-4
    keep_ref = s[:, 0] < 0.8 * s[:, 1]
    keep_gpu = np.sqrt(d2[:, 0].astype(np.float64)) < 0.8 * np.sqrt(d2[:, 1].astype(np.float64))
    np.testing.assert_array_equal(keep_gpu[margin & clear], keep_ref[margin & clear])


def test_edge_cases():
    rng = np.random.default_rng(1)
    d = _sift_like(rng, 6); kp = rng.random((6, 2)).astype(np.float32)
    m = D.DescriptorRatioMatcher(0.8, DEV)
    # pair 0: one train row (no second neighbour) -> nothing; pair 1: no query rows; pair 2: identical sets
    out = m([(kp, d), (kp[:0], d[:0]), (kp, d)], [(kp[:1], d[:1]), (kp, d), (kp, d)])
    assert out["n_corr"].cpu().tolist() == [0, 0, 6]
    np.testing.assert_array_equal(out["pts0"][2, :6].cpu().numpy(), kp)
    np.testing.assert_array_equal(out["pts1"][2, :6].cpu().numpy(), kp)
    assert out["nn_idx"][2, :6].cpu().tolist() == list(range(6))


def test_plugin_classes_with_supplied_detector():
    """SIFTMatching / SIFT_matcher with a caller-supplied detectAndCompute (OpenCV is absent offline)"""
    from mapfree_reloc_amd.config.default import cfg as default_cfg
    from mapfree_reloc_amd.matching.feature_matching import SIFTMatching
    from mapfree_reloc_amd.matchers import SIFT_matcher
    rng = np.random.default_rng(5)
    d0 = _sift_like(rng, 400); d1 = _sift_like(rng, 380)
    d1[:200] = np.clip(np.rint(d0[100:300] + rng.normal(0, 4, (200, 128))), 0, 255)
    kp0 = (rng.random((400, 2)) * 500).astype(np.float32); kp1 = (rng.random((380, 2)) * 500).astype(np.float32)
    feats = {0: (kp0, d0), 1: (kp1, d1)}

    def detector(gray):
        return feats[int(gray[0, 0])]

    cfg = default_cfg.clone()
    cfg.SIFT.NUM_FEATURES = 2048; cfg.SIFT.RATIO_THRESHOLD = 0.8; cfg.DEBUG = False
    sm = SIFTMatching(cfg, detector=detector)
    im0 = torch.zeros(1, 3, 16, 16); im1 = torch.full((1, 3, 16, 16), 1.0 / 255.0 + 1e-4)
    pts1, pts2 = sm.get_correspondences({"image0": im0, "image1": im1})
    ref0, ref1 = O.sift_ratio_match(d0, d1, kp0, kp1, 0.8)
    assert len(ref0) >= 190
    np.testing.assert_array_equal(pts1, ref0); np.testing.assert_array_equal(pts2, ref1)
    off = SIFT_matcher((16, 16), detector=detector)
    pts = off.match_arrays(np.zeros((16, 16), np.uint8), np.ones((16, 16), np.uint8))
    np.testing.assert_array_equal(pts, np.concatenate([ref0, ref1], 1))
