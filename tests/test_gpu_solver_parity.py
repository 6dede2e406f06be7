"""-m gpu parity tests: HIP pose-solver kernels (through the C-ABI) vs the CPU oracle on the
same seeded inputs.  Bar (BASELINE.json north_star): bit-exact inlier indices for a fixed
RANSAC seed, pose within 1e-4 rad / 1e-4 m (we assert far tighter: <= 1e-9)."""
import os

import numpy as np
import pytest
import torch

import mapfree_reloc_amd as mfr
from mapfree_reloc_amd import solver_ops as ops
from mapfree_reloc_amd import synth
from oracle import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_library_is_native_gfx950():
    lib = mfr._lib.load(require_gpu=True)
    assert lib.mfr_target_arch() == b"gfx950"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_f64_ieee_contract():
    """host and device agree bit-for-bit on / sqrt and unfused a*b+c (the FP contract of geom_dev.h)"""
    rng = np.random.default_rng(0)
    n = 1 << 18
    a = rng.normal(size=n) * 10.0 ** rng.uniform(-8, 8, n)
    b = rng.normal(size=n) * 10.0 ** rng.uniform(-8, 8, n)
    c = rng.normal(size=n) * 10.0 ** rng.uniform(-8, 8, n)
    out = ops.test_f64_ops(_dev(a), _dev(b), _dev(c)).cpu().numpy()
    np.testing.assert_array_equal(out[:, 0], a / b)
    np.testing.assert_array_equal(out[:, 1], np.sqrt(np.abs(a)))
    np.testing.assert_array_equal(out[:, 2], a * b + c)


@pytest.mark.parametrize("k", [4, 5])
def test_sampler_bit_exact(k):
    pair_ids = np.array([0, 7, 123456789012], dtype=np.int64)
    iters, n, seed = 300, 977, 42
    got = ops.test_sample(seed, _dev(pair_ids), iters, n, k).cpu().numpy()
    for b, pid in enumerate(pair_ids):
        for it in range(iters):
            np.testing.assert_array_equal(got[b, it], O.sample_distinct(seed, int(pid), it, n, k))


def test_pnp_lift_bit_exact():
    n_list = [300, 1024, 3, 0, 57]
    batch = synth.make_batch([11, 12, 13, 14, 15], n_list, maxN=1024, zero_depth_frac=0.2)
    xyz, obs, src, nv = ops.pnp_lift(_dev(batch["pts0"]), _dev(batch["pts1"]), _dev(batch["n_corr"]),
                                     _dev(batch["depth0"]), _dev(batch["K0"]))
    xyz, obs, src, nv = xyz.cpu().numpy(), obs.cpu().numpy(), src.cpu().numpy(), nv.cpu().numpy()
    for b, n in enumerate(n_list):
        rx, ro, rs = O.pnp_lift(batch["pts0"][b, :n], batch["pts1"][b, :n], batch["depth0"][b], batch["K0"][b])
        assert nv[b] == len(rx)
        np.testing.assert_array_equal(xyz[b, :nv[b]], rx)
        np.testing.assert_array_equal(obs[b, :nv[b]], ro)
        np.testing.assert_array_equal(src[b, :nv[b]], rs)


def _ransac_case(n_list, seeds, outl, iters=1000, seed=0, noise=1.0):
    batch = synth.make_batch(seeds, n_list, maxN=max(max(n_list), 4), outlier_frac=outl, noise_px=noise)
    xyz, obs, src, nv = ops.pnp_lift(_dev(batch["pts0"]), _dev(batch["pts1"]), _dev(batch["n_corr"]),
                                     _dev(batch["depth0"]), _dev(batch["K0"]))
    r = ops.pnp_ransac(xyz, obs, nv, _dev(batch["K1"]), _dev(batch["pair_ids"]), max_iters=iters, seed=seed)
    r = {k: v.cpu().numpy() for k, v in r.items()}
    xyz, obs, nv = xyz.cpu().numpy(), obs.cpu().numpy(), nv.cpu().numpy()
    for b in range(len(n_list)):
        m = int(nv[b])
        ref = O.pnp_ransac(xyz[b, :m], obs[b, :m], batch["K1"][b], max_iters=iters, seed=seed,
                           pair_id=int(batch["pair_ids"][b]), want_counts=True)
        assert r["status"][b] == ref["status"], (b, r["status"][b], ref["status"])
        if m > 4:
            run = ref["iters_run"]
            np.testing.assert_array_equal(r["counts"][b, :run], ref["counts"][:run])
            assert r["best_iter"][b] == ref["best_iter"]
            assert r["iters_run"][b] == ref["iters_run"]
        assert r["n_inliers"][b] == ref["n_inl"]
        if ref["status"] == 0:
            np.testing.assert_array_equal(r["mask"][b, :m], ref["mask"])          # bit-exact inlier indices
            np.testing.assert_array_equal(r["R"][b], ref["R"])                    # LM is bit-reproducible too
            np.testing.assert_array_equal(r["t"][b], ref["t"])
        else:
            assert np.isnan(r["R"][b]).all() and np.isnan(r["t"][b]).all()
    return batch, r


def test_pnp_ransac_bit_exact_vs_oracle():
    _ransac_case([256, 1024, 64, 900, 5, 4, 3, 0], [1, 2, 3, 4, 5, 6, 7, 8], outl=0.3)


def test_pnp_ransac_bit_exact_heavy_outliers_and_seeds():
    for seed in (0, 1, 99):
        _ransac_case([1024, 333, 2000], [21, 22, 23], outl=0.6, seed=seed)


def test_pnp_ransac_short_iteration_budget():
    _ransac_case([200, 500], [31, 32], outl=0.5, iters=37)
    _ransac_case([200], [33], outl=0.2, iters=1)


def test_pnp_known_answer_pose():
    batch, r = _ransac_case([1024, 512], [41, 42], outl=0.3, noise=0.5)
    for b in range(2):
        assert r["status"][b] == 0
        assert synth.rot_err_deg(r["R"][b], batch["R_gt"][b]) < 0.1
        assert np.linalg.norm(r["t"][b] - batch["t_gt"][b]) < 0.02


def test_pnp_solve_batch_end_to_end():
    n_list = [1024, 500, 3, 0, 4, 40, 777, 20]
    batch = synth.make_batch(list(range(50, 58)), n_list, maxN=1024, outlier_frac=0.4, zero_depth_frac=0.05)
    # pair 5: no valid depth at all -> BAD_DEPTH (pose_solver.py:197-198)
    batch["depth0"][5][:] = 0.0
    solver = ops.PnPBatchSolver(max_iters=1000, reproj_thr=3.0, confidence=0.9999, seed=3)
    out = solver(_dev(batch["pts0"]), _dev(batch["pts1"]), _dev(batch["n_corr"]), _dev(batch["depth0"]),
                 _dev(batch["K0"]), _dev(batch["K1"]), _dev(batch["pair_ids"]), want_mask=True)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b, n in enumerate(n_list):
        st, R, t, ninl = O.pnp_solve(batch["pts0"][b, :n], batch["pts1"][b, :n], batch["depth0"][b],
                                     batch["K0"][b], batch["K1"][b], seed=3, pair_id=int(batch["pair_ids"][b]))
        assert out["status"][b] == st, (b, out["status"][b], st)
        assert out["n_inliers"][b] == ninl
        assert out["mask"][b].sum() == ninl
        if st == 0:
            np.testing.assert_array_equal(out["R"][b], R)
            np.testing.assert_array_equal(out["t"][b], t.reshape(3))
        else:
            assert np.isnan(out["R"][b]).all() and np.isnan(out["t"][b]).all()
    assert out["status"][2] == ops.ST_TOO_FEW and out["status"][3] == ops.ST_TOO_FEW
    assert out["status"][5] == ops.ST_BAD_DEPTH


def test_scale_from_depth_vs_oracle_and_golden(golden_dir):
    # (a) fixtures produced by the reference's own python (pose_solver.py:137-172)
    g = np.load(os.path.join(golden_dir, "ref_emat_metric.npz"))
    nc = int(g["n_cases"])
    maxN = max(len(g[f"c{c}_pts0"]) for c in range(nc))
    B = nc
    H, W = g["c0_depth0"].shape
    pts0 = np.zeros((B, maxN, 2), np.float32); pts1 = np.zeros((B, maxN, 2), np.float32)
    mask = np.zeros((B, maxN), np.uint8); n_corr = np.zeros(B, np.int32)
    for c in range(nc):
        n = len(g[f"c{c}_pts0"]); n_corr[c] = n
        pts0[c, :n] = g[f"c{c}_pts0"]; pts1[c, :n] = g[f"c{c}_pts1"]; mask[c, :n] = g[f"c{c}_mask"]
    st = lambda k: np.stack([g[f"c{c}_{k}"] for c in range(nc)])
    solver = ops.ScaleFromDepthBatch(0.1)
    out = solver(_dev(pts0), _dev(pts1), _dev(mask), _dev(n_corr), _dev(st("depth0")), _dev(st("depth1")),
                 _dev(st("K0")), _dev(st("K1")), _dev(st("R_in")), _dev(st("t_in")))
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for c in range(nc):
        ref_inl = int(g[f"c{c}_inliers"])
        assert out["n_inliers"][c] == ref_inl
        if ref_inl == 0:
            assert out["status"][c] == ops.ST_BAD_DEPTH and np.isnan(out["t_metric"][c]).all()
        else:
            np.testing.assert_allclose(out["t_metric"][c], g[f"c{c}_t_out"], rtol=1e-6, atol=1e-9)
            # and bit-exact vs the oracle
            n = n_corr[c]
            sc = O.scale_lift(pts0[c, :n], pts1[c, :n], mask[c, :n], st("depth0")[c], st("depth1")[c],
                              st("K0")[c], st("K1")[c], st("R_in")[c], st("t_in")[c])
            cnt, bs, _ = O.scale_ransac(sc, 0.1)
            assert cnt == out["n_inliers"][c] and bs == out["best_scale"][c]


def test_scale_from_depth_full_size_bit_exact():
    n_list = [1024, 3000, 7, 0]
    batch = synth.make_batch([61, 62, 63, 64], n_list, maxN=3000, outlier_frac=0.3, depth_noise=0.05,
                             zero_depth_frac=0.03)
    rng = np.random.default_rng(5)
    mask = (rng.uniform(size=(4, 3000)) < 0.8).astype(np.uint8)
    R = batch["R_gt"]; t = batch["t_gt"] / np.linalg.norm(batch["t_gt"], axis=1, keepdims=True)
    in_status = np.array([0, 0, 0, 3], np.int32)
    solver = ops.ScaleFromDepthBatch(0.1)
    out = solver(_dev(batch["pts0"]), _dev(batch["pts1"]), _dev(mask), _dev(batch["n_corr"]), _dev(batch["depth0"]),
                 _dev(batch["depth1"]), _dev(batch["K0"]), _dev(batch["K1"]), _dev(R), _dev(t), _dev(in_status))
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for b, n in enumerate(n_list[:3]):
        sc = O.scale_lift(batch["pts0"][b, :n], batch["pts1"][b, :n], mask[b, :n], batch["depth0"][b],
                          batch["depth1"][b], batch["K0"][b], batch["K1"][b], R[b], t[b])
        cnt, bs, _ = O.scale_ransac(sc, 0.1)
        assert out["n_inliers"][b] == cnt
        assert out["best_scale"][b] == bs
        np.testing.assert_array_equal(out["t_metric"][b], bs * t[b])
    # known answer: recovered scale ~ |t_gt|
    assert abs(out["best_scale"][0] - np.linalg.norm(batch["t_gt"][0])) < 0.1
    assert out["status"][3] == 3 and out["n_inliers"][3] == 0
