"""-m gpu parity: E-matrix RANSAC path (csrc/emat.hip) vs the CPU oracle (oracle/mfr_oracle_emat.c),
and the full EssentialMatrixMetric chain, for both model-quality methods: MAGSAC++ loss + sigma-consensus++ (the default; what
the reference asks OpenCV for, pose_solver.py:46-48) and the inlier count + LM polish of rounds 1-3.  Bit-exact hypothesis
losses / counts, selected iteration, iteration budget, number of local optimisations, inlier masks and (because reductions are
wave64-ordered on both sides) poses."""
import numpy as np
import pytest
import torch

from mapfree_reloc_amd import solver_ops as ops
from mapfree_reloc_amd import synth
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


SCORES = {"magsac": O.EMAT_MAGSAC, "count": O.EMAT_COUNT}


def test_magsac_table_is_the_oracles():
    """the one place libm enters (erfc, exp at table-fill time): the library's host table and the oracle's are the same bits"""
    np.testing.assert_array_equal(ops.magsac_lut_host(), O.magsac_lut())


def _case(n_list, seeds, outl, iters=1000, seed=0, thr=2.0, score="magsac", ratio=1.0, noise=1.0):
    batch = synth.make_batch(seeds, n_list, maxN=max(max(n_list), 8), outlier_frac=outl, noise_px=noise)
    solver = ops.EssentialBatchSolver(thr, 0.9999, seed, iters, score=score, max_thr_ratio=ratio)
    out = solver(_dev(batch["pts0"]), _dev(batch["pts1"]), _dev(batch["n_corr"]), _dev(batch["K0"]), _dev(batch["K1"]),
                 _dev(batch["pair_ids"]), diagnostics=True)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b, n in enumerate(n_list):
        ref = O.emat_solve(batch["pts0"][b, :n], batch["pts1"][b, :n], batch["K0"][b], batch["K1"][b], thr, 0.9999, iters,
                           seed, int(batch["pair_ids"][b]), want_counts=True, score=SCORES[score], max_thr_ratio=ratio)
        assert out["status"][b] == ref["status"], (b, out["status"][b], ref["status"])
        if n > 5:
            run = ref["iters_run"]
            np.testing.assert_array_equal(out["counts"][b, :run], ref["counts"][:run])
            if score == "magsac":
                np.testing.assert_array_equal(out["losses"][b, :run], ref["losses"][:run])
                assert out["lo_runs"][b] == ref["lo_runs"]
            assert out["best_iter"][b] == ref["best_iter"] and out["iters_run"][b] == ref["iters_run"]
        assert out["n_inliers"][b] == ref["n_inl"]
        np.testing.assert_array_equal(out["mask"][b, :n], ref["mask"])
        if ref["status"] == 0:
            np.testing.assert_array_equal(out["R"][b], ref["R"])
            np.testing.assert_array_equal(out["t"][b], ref["t"])
            assert np.abs(out["R"][b].T @ out["R"][b] - np.eye(3)).max() < 1e-14      # a rotation, not Horn's near-rotation
        else:
            assert np.isnan(out["R"][b]).all()
    return batch, out


@pytest.mark.parametrize("score", ["magsac", "count"])
def test_emat_bit_exact_vs_oracle(score):
    _case([256, 1024, 64, 5, 4, 0, 7, 2500], [1, 2, 3, 4, 5, 6, 7, 8], outl=0.3, score=score)


@pytest.mark.parametrize("score", ["magsac", "count"])
def test_emat_heavy_outliers_seeds_and_short_budget(score):
    _case([1024, 400], [11, 12], outl=0.6, seed=5, score=score)
    _case([300, 800], [13, 14], outl=0.5, iters=41, score=score)


def test_emat_magsac_local_optimisation_inside_the_loop_and_wide_sigma():
    """70 % outliers: new best models keep arriving after iteration 100, so sigma-consensus++ runs INSIDE the replay (lo_runs > 1)
    and its result changes which later hypotheses become the best; max_thr_ratio 2: the loss cut and the tentative-inlier
    threshold are different sets"""
    _, out = _case([900, 1500, 3000], [41, 42, 43], outl=0.7, score="magsac")
    assert (out["lo_runs"] > 1).any()
    _case([700, 1100], [44, 45], outl=0.5, score="magsac", ratio=2.0)
    _case([700, 1100], [46, 47], outl=0.4, score="magsac", ratio=1.5, noise=0.3, thr=1.0)


@pytest.mark.parametrize("score", ["magsac", "count"])
def test_emat_known_answer_pose(score):
    batch, out = _case([2000, 800], [21, 22], outl=0.3, score=score)
    for b in range(2):
        tg = batch["t_gt"][b] / np.linalg.norm(batch["t_gt"][b])
        assert synth.rot_err_deg(out["R"][b], batch["R_gt"][b]) < 0.5
        assert np.degrees(np.arccos(np.clip(out["t"][b] @ tg, -1, 1))) < 3.0


def test_emat_metric_chain_vs_oracle():
    """EssentialMatrixMetricSolver = E-mat -> scale from depth (pose_solver.py:125-172)"""
    n_list = [1500, 600, 3]
    batch = synth.make_batch([31, 32, 33], n_list, maxN=1500, outlier_frac=0.3, depth_noise=0.02)
    em = ops.EssentialBatchSolver(2.0, 0.9999, 0)
    sc = ops.ScaleFromDepthBatch(0.1)
    d = {k: _dev(v) for k, v in batch.items() if isinstance(v, np.ndarray)}
    e = em(d["pts0"], d["pts1"], d["n_corr"], d["K0"], d["K1"], d["pair_ids"])
    s = sc(d["pts0"], d["pts1"], e["mask"], d["n_corr"], d["depth0"], d["depth1"], d["K0"], d["K1"], e["R"], e["t"], e["status"])
    e = {k: v.cpu().numpy() for k, v in e.items()}; s = {k: v.cpu().numpy() for k, v in s.items()}
    for b, n in enumerate(n_list):
        ref = O.emat_solve(batch["pts0"][b, :n], batch["pts1"][b, :n], batch["K0"][b], batch["K1"][b], 2.0, 0.9999, 1000, 0,
                           int(batch["pair_ids"][b]))
        if ref["status"] != 0:
            assert s["status"][b] != 0 and np.isnan(s["t_metric"][b]).all() and s["n_inliers"][b] == 0
            continue
        scale = O.scale_lift(batch["pts0"][b, :n], batch["pts1"][b, :n], ref["mask"], batch["depth0"][b], batch["depth1"][b],
                             batch["K0"][b], batch["K1"][b], ref["R"], ref["t"])
        cnt, bs, _ = O.scale_ransac(scale, 0.1)
        assert s["n_inliers"][b] == cnt and s["best_scale"][b] == bs
        np.testing.assert_array_equal(s["t_metric"][b], bs * ref["t"])
        # known answer: metric translation within 5 cm / 10 % of the truth
        assert np.linalg.norm(s["t_metric"][b] - batch["t_gt"][b]) < max(0.05, 0.1 * np.linalg.norm(batch["t_gt"][b]))
