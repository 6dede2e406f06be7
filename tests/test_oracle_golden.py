"""Oracle (oracle/mfr_oracle.c) vs fixtures produced by executing the reference's own
Python (oracle/gen_golden.py; reference lib/models/matching/pose_solver.py,
feature_matching.py, etc/feature_matching_baselines/utils.py)."""
import os

import numpy as np

from oracle import oracle_lib as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_backproject_matches_reference(golden_dir):
    g = _load(golden_dir, "ref_backproject.npz")
    for c in range(int(g["n_cases"])):
        xyz = O.backproject(g[f"c{c}_uv"], g[f"c{c}_d"], g[f"c{c}_K"])
        ref = g[f"c{c}_xyz"]
        # f32 inverse of K may differ by 1 ulp(f32) from LAPACK's; products are f64
        np.testing.assert_allclose(xyz, ref, rtol=3e-7, atol=1e-12)


def test_pnp_lift_matches_reference(golden_dir):
    g = _load(golden_dir, "ref_pnp_lift.npz")
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        pts0, pts1 = g[p + "pts0"], g[p + "pts1"]
        if len(pts0) < 4:
            assert int(g[p + "called"]) == 0
            continue
        xyz, obs, src = O.pnp_lift(pts0, pts1, g[p + "depth0"], g[p + "K0"])
        if int(g[p + "called"]) == 0:
            assert len(xyz) < 4        # pose_solver.py:197-198
            continue
        ref_xyz, ref_obs = g[p + "xyz"], g[p + "obs"]
        assert xyz.shape == ref_xyz.shape
        np.testing.assert_allclose(xyz, ref_xyz, rtol=3e-7, atol=1e-12)
        np.testing.assert_array_equal(obs, ref_obs)
        np.testing.assert_array_equal(obs, pts1[src].astype(np.float64))


def test_emat_metric_scale_matches_reference(golden_dir):
    g = _load(golden_dir, "ref_emat_metric.npz")
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        R, t = g[p + "R_in"], g[p + "t_in"]
        scale = O.scale_lift(g[p + "pts0"], g[p + "pts1"], g[p + "mask"], g[p + "depth0"], g[p + "depth1"],
                             g[p + "K0"], g[p + "K1"], R, t)
        ref_inl = int(g[p + "inliers"])
        if len(scale) < 1:
            assert ref_inl == 0 and np.isnan(g[p + "R_out"]).all()     # pose_solver.py:145-149
            continue
        n, best_scale, _ = O.scale_ransac(scale, 0.1)
        assert n == ref_inl
        np.testing.assert_allclose(best_scale * t, g[p + "t_out"], rtol=1e-6, atol=1e-9)
        np.testing.assert_array_equal(R, g[p + "R_out"])


def test_wire_format_roundtrip_fixture(golden_dir):
    g = _load(golden_dir, "ref_wire_format.npz")
    stack = g["stack"]
    assert stack.dtype == np.float64 and stack.shape == (6, 33, 4)
    for i in range(int(g["n_pairs"])):
        row = stack[i].astype(np.float32)
        row = row[~np.isnan(row)].reshape(-1, 4)
        if len(row):
            np.testing.assert_array_equal(row[:, :2], g[f"p1_{i}"])
            np.testing.assert_array_equal(row[:, 2:], g[f"p2_{i}"])
        else:
            assert g[f"p1_{i}"].shape == (0,)


def test_rootsift_matches_reference_bitwise(golden_dir):
    """root_sift (feature_matching.py:68-74) executed by the reference's own numpy code"""
    g = _load(golden_dir, "ref_sift_ratio.npz")
    for name in ("rs_int", "rs_float"):
        out, n2 = O.rootsift(g[name + "_in"])
        np.testing.assert_array_equal(out, g[name + "_out"])
        np.testing.assert_allclose(n2, (out.astype(np.float64) ** 2).sum(1), rtol=2e-6)


def test_sift_ratio_loop_matches_reference(golden_dir):
    """SIFTMatching.get_correspondences (feature_matching.py:75-118) run with stubbed detection and an
    exact 2-NN in place of FLANN: same kept matches, same order, same coordinates"""
    g = _load(golden_dir, "ref_sift_ratio.npz")
    for ci in range(3):
        p0, p1 = O.sift_ratio_match(g[f"gc{ci}_des0"].astype(np.float32), g[f"gc{ci}_des1"].astype(np.float32),
                                    g[f"gc{ci}_kp0"], g[f"gc{ci}_kp1"], 0.8)
        assert len(p0) > 0
        np.testing.assert_array_equal(p0, g[f"gc{ci}_pts1"])
        np.testing.assert_array_equal(p1, g[f"gc{ci}_pts2"])


def test_sift_ratio_edge_cases():
    rng = np.random.default_rng(3)
    d = (rng.random((5, 128)) * 100).astype(np.float32)
    kp = rng.random((5, 2)).astype(np.float32)
    # a single train descriptor: knnMatch(k=2) has no second neighbour -> no correspondences
    p0, p1 = O.sift_ratio_match(d, d[:1], kp, kp[:1])
    assert p0.shape == (0, 2) and p1.shape == (0, 2)
    # duplicate train rows: d1 == d2 == 0 -> 0 < 0.8*0 is False -> rejected
    p0, _ = O.sift_ratio_match(d, np.concatenate([d, d]), kp, np.concatenate([kp, kp]))
    assert len(p0) == 0
    # identical sets: best distance 0, second > 0 -> every query kept, identity assignment
    p0, p1 = O.sift_ratio_match(d, d, kp, kp)
    np.testing.assert_array_equal(p0, kp); np.testing.assert_array_equal(p1, kp)


# ---------------------------------------------------------------------------------------------------------------------
# float64 intrinsics: the flow of the real Map-free loader (lib/datasets/utils.py:117-130 multiplies a float64 eye(3) into
# K; lib/datasets/mapfree.py:50-52 always calls it).  Fixture = the reference's own read_intrinsics -> backproject_3d /
# EssentialMatrix(Metric)Solver / PnPSolver executed with that K (oracle/gen_k64_golden.py).  Everything BIT-EXACT.
# ---------------------------------------------------------------------------------------------------------------------
def _k64_cases(golden_dir):
    g = _load(golden_dir, "ref_k64.npz")
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        d = {k[len(p):]: g[k] for k in g.files if k.startswith(p)}
        d["depth0"] = d["depth0"].astype(np.float32) / 1000      # lib/datasets/utils.py:77-81
        d["depth1"] = d["depth1"].astype(np.float32) / 1000
        yield c, d


def test_k64_fixture_holds_both_dtype_flows(golden_dir):
    kinds = [d["K0"].dtype for _, d in _k64_cases(golden_dir)]
    assert kinds.count(np.dtype(np.float64)) >= 9 and kinds.count(np.dtype(np.float32)) >= 3


def test_k64_backproject_bit_exact(golden_dir):
    for c, d in _k64_cases(golden_dir):
        uv = np.int32(d["pts0"])
        xyz = O.backproject(uv, d["depth0"][uv[:, 1], uv[:, 0]], d["K0"])
        np.testing.assert_array_equal(xyz, d["bp_xyz"], err_msg=f"case {c} K dtype {d['K0'].dtype}")


def test_k64_emat_normalisation_and_threshold_bit_exact(golden_dir):
    """pose_solver.py:39-43 in K's own dtype: what the reference hands to cv.findEssentialMat"""
    for c, d in _k64_cases(golden_dir):
        np.testing.assert_array_equal(O.normalize_points(d["pts0"], d["K0"]), d["k0n"], err_msg=f"case {c}")
        np.testing.assert_array_equal(O.normalize_points(d["pts1"], d["K1"]), d["k1n"], err_msg=f"case {c}")
        thr = O.emat_threshold(2.0, d["K0"], d["K1"])
        if d["K0"].dtype == np.float64:
            assert thr == float(d["thr"]), c
        else:
            # float32 K: `PIX_THRESHOLD / np.float32` is a float64 division under the reference's pinned numpy 1.24 (what the
            # oracle restates) and a float32 one under NEP 50 / numpy 2.x (what generated the fixture): equal to f32 round-off
            assert abs(thr - float(d["thr"])) <= 6e-8 * thr, c


def test_k64_scale_from_depth_bit_exact(golden_dir):
    for c, d in _k64_cases(golden_dir):
        scale = O.scale_lift(d["pts0"], d["pts1"], d["mask"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["R_in"], d["t_in"])
        if len(scale) < 1:
            assert int(d["inliers"]) == 0
            continue
        n, best_scale, _ = O.scale_ransac(scale, 0.1)
        assert n == int(d["inliers"]), c
        # the count (= submission confidence) is exact; the scale itself went through numpy's BLAS (`R @ xyz0.T`, `np.dot(.., t)`:
        # FMA kernels, machine-dependent summation) in the reference and through unfused IEEE ops here: a few ulp
        np.testing.assert_allclose(best_scale * d["t_in"], d["t_out"], rtol=1e-13, atol=1e-15, err_msg=f"case {c}")


def test_k64_pnp_lift_bit_exact(golden_dir):
    for c, d in _k64_cases(golden_dir):
        xyz, obs, src = O.pnp_lift(d["pts0"], d["pts1"], d["depth0"], d["K0"])
        np.testing.assert_array_equal(xyz, d["pnp_xyz"], err_msg=f"case {c}")
        np.testing.assert_array_equal(obs, d["pnp_obs"])
        np.testing.assert_array_equal(np.asarray(d["K1"], np.float64), d["pnp_K"])     # K handed to OpenCV: exact widening
