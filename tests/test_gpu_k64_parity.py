"""-m gpu: the FLOAT64-intrinsics flow of the real Map-free loader through the C-ABI (k_dtype = MFR_K_F64).

`correct_intrinsic_scale` (lib/datasets/utils.py:117-130) multiplies a float64 eye(3) into K and Map-free always calls it
(lib/datasets/mapfree.py:50-52), so `data['K_color*']` is float64 and the reference evaluates np.linalg.inv(K)
(pose_solver.py:16), the K-normalisation (:39-40) and the threshold mean (:43) in float64.  Checked here:
  (1) HIP == the reference's own Python (tests/golden/ref_k64.npz, oracle/gen_k64_golden.py) BIT FOR BIT on every
      reference-owned line: PnP lift, E-mat normalisation -> the RANSAC sees the same x / threshold (observed through the
      hypothesis counts vs the oracle, which is itself pinned on k0n / k1n / thr), scale-from-depth count;
  (2) HIP == CPU oracle bit for bit for the whole solvers (PnP, E-mat + scale, Procrustes + ICP) with float64 K;
  (3) the dtype matters: the float32 flow on the same values gives different bits (so a silent cast would be seen);
  (4) the per-pair plugins pass float64 K through untouched.
"""
import os

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


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_k64.npz"))
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        d = {k[len(p):]: g[k] for k in g.files if k.startswith(p)}
        d["depth0"] = d["depth0"].astype(np.float32) / 1000
        d["depth1"] = d["depth1"].astype(np.float32) / 1000
        yield c, d


def test_pnp_lift_equals_reference_python_bitwise(golden_dir):
    for c, d in _cases(golden_dir):
        n = len(d["pts0"])
        xyz, obs, src, nv = ops.pnp_lift(_dev(d["pts0"][None]), _dev(d["pts1"][None]), _dev(np.array([n], np.int32)),
                                         _dev(d["depth0"][None]), _dev(d["K0"][None]))
        m = int(nv[0])
        assert m == len(d["pnp_xyz"]), c
        np.testing.assert_array_equal(xyz[0, :m].cpu().numpy(), d["pnp_xyz"], err_msg=f"case {c} ({d['K0'].dtype})")
        np.testing.assert_array_equal(obs[0, :m].cpu().numpy(), d["pnp_obs"])


def test_scale_from_depth_equals_reference_python(golden_dir):
    """EssentialMatrixMetricSolver's own lines (pose_solver.py:137-172) with the reference's float64 K: the consensus count
    (= the submission confidence) equals the reference's, the metric translation agrees to BLAS round-off"""
    sc = ops.ScaleFromDepthBatch(0.1)
    for c, d in _cases(golden_dir):
        n = len(d["pts0"])
        out = sc(_dev(d["pts0"][None]), _dev(d["pts1"][None]), _dev(d["mask"][None].astype(np.uint8)), _dev(np.array([n], np.int32)),
                 _dev(d["depth0"][None]), _dev(d["depth1"][None]), _dev(d["K0"][None]), _dev(d["K1"][None]),
                 _dev(d["R_in"][None]), _dev(d["t_in"][None]))
        assert int(out["n_inliers"][0]) == int(d["inliers"]), c
        if int(d["inliers"]) > 0:
            np.testing.assert_allclose(out["t_metric"][0].cpu().numpy(), d["t_out"], rtol=1e-13, atol=1e-15)
            # and bit-exact against the oracle (unfused IEEE on both sides)
            s = O.scale_lift(d["pts0"], d["pts1"], d["mask"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["R_in"], d["t_in"])
            cnt, bs, _ = O.scale_ransac(s, 0.1)
            assert cnt == int(out["n_inliers"][0]) and bs == float(out["best_scale"][0])


@pytest.mark.parametrize("kdt", [np.float64, np.float32])
def test_solvers_bit_exact_vs_oracle_in_both_k_dtypes(kdt):
    """whole solvers, same correspondences, K in float64 (Map-free) and float32 (resize=None): HIP == oracle bit for bit"""
    n_list = [700, 300, 64, 5]
    batch = synth.make_batch([101, 102, 103, 104], n_list, maxN=1024, outlier_frac=0.35)
    K0 = batch["K0"].astype(kdt); K1 = batch["K1"].astype(kdt)
    if kdt is np.float64:                     # values a float32 cannot hold, as a rescaled Map-free K has
        K0 = K0 * (1.0 + 1e-9); K1 = K1 * (1.0 - 3e-10)
        K0[:, 2, 2] = 1.0; K1[:, 2, 2] = 1.0
    dv = {k: _dev(batch[k]) for k in ("pts0", "pts1", "n_corr", "depth0", "depth1", "pair_ids")}
    dK0, dK1 = _dev(K0), _dev(K1)
    pnp = ops.PnPBatchSolver(1000, 3.0, 0.9999, seed=0)(dv["pts0"], dv["pts1"], dv["n_corr"], dv["depth0"], dK0, dK1, dv["pair_ids"], want_mask=True)
    em = ops.EssentialBatchSolver(2.0, 0.9999, seed=0)(dv["pts0"], dv["pts1"], dv["n_corr"], dK0, dK1, dv["pair_ids"], diagnostics=True)
    sc = ops.ScaleFromDepthBatch(0.1)(dv["pts0"], dv["pts1"], em["mask"], dv["n_corr"], dv["depth0"], dv["depth1"], dK0, dK1, em["R"], em["t"], em["status"])
    pr = ops.ProcrustesBatchSolver(0.05, 0.999, seed=0)(dv["pts0"], dv["pts1"], dv["n_corr"], dv["depth0"], dv["depth1"], dK0, dK1, dv["pair_ids"])
    torch.cuda.synchronize()
    for b, n in enumerate(n_list):
        a = dict(pts0=batch["pts0"][b, :n], pts1=batch["pts1"][b, :n], d0=batch["depth0"][b], d1=batch["depth1"][b])
        pid = int(batch["pair_ids"][b])
        st, R, t, ninl = O.pnp_solve(a["pts0"], a["pts1"], a["d0"], K0[b], K1[b], seed=0, pair_id=pid)
        assert int(pnp["status"][b]) == st and int(pnp["n_inliers"][b]) == ninl, (b, kdt)
        if st == 0:
            np.testing.assert_array_equal(pnp["R"][b].cpu().numpy(), R); np.testing.assert_array_equal(pnp["t"][b].cpu().numpy(), t.reshape(3))
        e = O.emat_solve(a["pts0"], a["pts1"], K0[b], K1[b], 2.0, 0.9999, 1000, seed=0, pair_id=pid, want_counts=True)
        assert int(em["status"][b]) == e["status"] and int(em["n_inliers"][b]) == e["n_inl"], (b, kdt)
        if n > 5:
            run = e["iters_run"]
            np.testing.assert_array_equal(em["counts"][b, :run].cpu().numpy(), e["counts"][:run])
        if e["status"] == 0:
            np.testing.assert_array_equal(em["mask"][b, :n].cpu().numpy(), e["mask"])
            np.testing.assert_array_equal(em["R"][b].cpu().numpy(), e["R"])
            s = O.scale_lift(a["pts0"], a["pts1"], e["mask"], a["d0"], a["d1"], K0[b], K1[b], e["R"], e["t"])
            if len(s):
                cnt, bs, _ = O.scale_ransac(s, 0.1)
                assert cnt == int(sc["n_inliers"][b]) and bs == float(sc["best_scale"][b])
        st, R, t, ninl = O.procrustes_solve(a["pts0"], a["pts1"], a["d0"], a["d1"], K0[b], K1[b], seed=0, pair_id=pid)
        assert int(pr["status"][b]) == st and int(pr["n_inliers"][b]) == ninl, (b, kdt)
        if st == 0:
            np.testing.assert_array_equal(pr["R"][b].cpu().numpy(), R)


def test_k_dtype_is_not_silently_cast():
    """the same VALUES as float32 and as float64 must give different lifted bits somewhere (f32 vs f64 inverse): proves the
    library really switches arithmetic on k_dtype and nothing upstream of it narrows a float64 K"""
    batch = synth.make_batch([7], [900], maxN=1024)
    args = [_dev(batch[k]) for k in ("pts0", "pts1", "n_corr", "depth0")]
    x32 = ops.pnp_lift(*args, _dev(batch["K0"].astype(np.float32)))[0].cpu().numpy()
    x64 = ops.pnp_lift(*args, _dev(batch["K0"].astype(np.float64)))[0].cpu().numpy()
    assert not np.array_equal(x32, x64)
    np.testing.assert_allclose(x32, x64, rtol=1e-6, atol=1e-6)
    with pytest.raises(TypeError):
        ops.pnp_lift(*args, _dev(batch["K0"].astype(np.float16)))


def test_icp_bit_exact_with_float64_K():
    H, W = 60, 80
    rng = np.random.default_rng(5)
    K = np.array([[70.0 * (1 + 1e-9), 0, 39.5], [0, 70.0, 29.5 + 1e-10], [0, 0, 1]], np.float64)
    z = 2.0 + 0.3 * np.sin(np.arange(W)[None] / 9.0) + 0.2 * np.cos(np.arange(H)[:, None] / 7.0)
    d0 = (np.round(z * 1000) / 1000).astype(np.float32)
    d1 = (np.round((z + 0.004) * 1000) / 1000).astype(np.float32)
    R0 = np.eye(3); t0 = np.array([0.002, -0.001, 0.003])
    ref = O.procrustes_icp(d0, d1, K, K, R0, t0)
    Rd, td = _dev(R0[None].copy()), _dev(t0[None].copy())
    out = ops.ProcrustesIcpRefine()(_dev(d0[None]), _dev(d1[None]), _dev(K[None]), _dev(K[None]), Rd, td)
    assert int(out["n_inliers"][0]) == ref["n_inliers"] and int(out["iters"][0]) == ref["iters"]
    np.testing.assert_array_equal(out["R"][0].cpu().numpy(), ref["R"])
    np.testing.assert_array_equal(out["t"][0].cpu().numpy(), ref["t"])


def test_plugins_pass_float64_K_through(golden_dir):
    """PnPSolver / EssentialMatrixMetricSolver plugins (batch-1 API) fed the fixture's float64 `data` dict == oracle on float64 K"""
    from mapfree_reloc_amd.config import get_cfg_defaults
    from mapfree_reloc_amd.matching import pose_solver as PS
    cfg = get_cfg_defaults()
    cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE = 1000, 3, 0.9999
    cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.SCALE_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE = 2.0, 0.1, 0.9999
    pnp, emm = PS.PnPSolver(cfg), PS.EssentialMatrixMetricSolver(cfg)
    seen64 = 0
    for c, d in _cases(golden_dir):
        if len(d["pts0"]) < 8:
            continue
        batch = synth.make_batch([500 + c], [400], maxN=400, outlier_frac=0.3)          # real geometry, the fixture's K dtype
        K0 = batch["K0"][0].astype(d["K0"].dtype); K1 = batch["K1"][0].astype(d["K0"].dtype)
        if K0.dtype == np.float64:
            K0[0, 0] *= (1 + 1e-9); K1[1, 1] *= (1 - 1e-9); seen64 += 1
        data = {"K_color0": torch.from_numpy(K0[None]), "K_color1": torch.from_numpy(K1[None]),
                "depth0": torch.from_numpy(batch["depth0"]), "depth1": torch.from_numpy(batch["depth1"]), "pair_id": torch.tensor([c])}
        p0, p1 = batch["pts0"][0], batch["pts1"][0]
        R, t, ninl = pnp.estimate_pose(p0, p1, data)
        st, Rr, tr, nr = O.pnp_solve(p0, p1, batch["depth0"][0], K0, K1, seed=0, pair_id=c)
        assert st == 0 and ninl == nr
        np.testing.assert_array_equal(R, Rr); np.testing.assert_array_equal(t, tr)
        R, t, ninl = emm.estimate_pose(p0, p1, data)
        e = O.emat_solve(p0, p1, K0, K1, cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE, 1000, seed=0, pair_id=c)
        s = O.scale_lift(p0, p1, e["mask"], batch["depth0"][0], batch["depth1"][0], K0, K1, e["R"], e["t"])
        cnt, bs, _ = O.scale_ransac(s, cfg.EMAT_RANSAC.SCALE_THRESHOLD)
        assert ninl == cnt
        np.testing.assert_array_equal(R, e["R"]); np.testing.assert_array_equal(t.reshape(3), bs * e["t"])
    assert seen64 >= 6
