"""Pins the restated OpenCV / Open3D arithmetic against fixtures produced by tests/external/gen_cv_golden.py (run once where
opencv-python 4.8 / open3d 0.17 exist).  Without the fixtures the tests are SKIPPED -- and DESIGN.md keeps saying "parity
unpinned vs OpenCV" for those rows.  The libraries draw different RANSAC samples (and OpenCV scores with MAGSAC++), so the
comparison is statistical, on the SURVEY.md 8d known-answer sets: pose within the resolution RANSAC itself has on 1 px noise,
inlier sets overlapping."""
import os

import numpy as np
import pytest

import mapfree_reloc_amd  # noqa: F401
from mapfree_reloc_amd import synth
from oracle import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} absent: run tests/external/gen_cv_golden.py where opencv-python / open3d are installed")
    return np.load(path, allow_pickle=False)


def _cases(z):
    return [(int(s), int(n), float(o)) for s, n, o in z["cases"]]


def _ang(Ra, Rb):
    return synth.rot_err_deg(np.asarray(Ra), np.asarray(Rb))


def test_oracle_pnp_vs_opencv_golden():
    z = _load("cv_pnp.npz")
    for seed, n, outl in _cases(z):
        p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
        st, R, t, ninl = O.pnp_solve(p["pts0"], p["pts1"], p["depth0"], p["K0"], p["K1"], 1000, 3.0, 0.9999, 0, seed)
        assert st == 0
        # both solvers vs each other and vs the ground truth: the restatement may not be worse than OpenCV by more than the
        # RANSAC-to-RANSAC spread on this noise level
        assert _ang(R, z[f"s{seed}_R"]) < 0.15 and np.linalg.norm(t.ravel() - z[f"s{seed}_t"]) < 0.02
        assert _ang(R, p["R_gt"]) < _ang(z[f"s{seed}_R"], p["R_gt"]) + 0.1
        assert abs(ninl - int(z[f"s{seed}_ninl"])) <= 0.03 * int(z[f"s{seed}_ninl"]) + 4


def test_oracle_emat_vs_opencv_golden():
    z = _load("cv_emat.npz")
    for seed, n, outl in _cases(z):
        p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
        out = O.emat_solve(p["pts0"], p["pts1"], p["K0"], p["K1"], 2.0, 0.9999, 1000, 0, seed)
        assert out["status"] == 0
        assert _ang(out["R"], z[f"s{seed}_R"]) < 0.3
        cosang = abs(float(out["t"].reshape(3) @ z[f"s{seed}_t"].reshape(3)))
        assert np.degrees(np.arccos(np.clip(cosang, -1, 1))) < 2.0
        m_cv = z[f"s{seed}_mask"].astype(bool)
        inter = (out["mask"].astype(bool) & m_cv).sum()
        assert inter >= 0.9 * min(m_cv.sum(), out["mask"].sum())            # MAGSAC++ vs inlier counting: near-identical consensus sets


def test_oracle_magsac_vs_opencv_model():
    """round 4: with OpenCV's own E on file (`s*_E`, `s*_sampson2`, `s*_mask_raw`), compare at the MODEL level, which does not
    depend on which samples each RANSAC drew: (i) OpenCV's raw mask is `sampson^2 < thr^2` of its E (what error / compare USAC
    uses); (ii) the oracle's winner has a MAGSAC++ loss no worse than OpenCV's model re-scored by the oracle (both went through
    a local optimisation, so neither should be beatable by the other's by more than the loss of a few points)"""
    z = _load("cv_emat.npz")
    if not any(k.endswith("_sampson2") for k in z.files):
        pytest.skip("cv_emat.npz predates round 4: regenerate with tests/external/gen_cv_golden.py")
    lut = O.magsac_lut()
    for seed, n, outl in _cases(z):
        p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
        thr2 = float(z[f"s{seed}_thr"]) ** 2
        r2 = z[f"s{seed}_sampson2"]
        raw = z[f"s{seed}_mask_raw"].astype(bool)
        assert ((r2 < thr2) == raw).mean() > 0.98                              # (i)
        out = O.emat_solve(p["pts0"], p["pts1"], p["K0"], p["K1"], 2.0, 0.9999, 1000, 0, seed)
        x0 = O.normalize_points(p["pts0"], p["K0"]); x1 = O.normalize_points(p["pts1"], p["K1"])

        def loss(E):
            pp0 = np.c_[x0, np.ones(len(x0))]; pp1 = np.c_[x1, np.ones(len(x1))]
            a = pp0 @ E.T; b = pp1 @ E
            q = np.sum(pp1 * a, 1) ** 2 / (a[:, 0] ** 2 + a[:, 1] ** 2 + b[:, 0] ** 2 + b[:, 1] ** 2)
            u = q[q < thr2] / thr2 * (len(lut) - 1)
            j = np.minimum(u.astype(int), len(lut) - 2)
            return float(np.sum(lut[j, 0] + (u - j) * (lut[j + 1, 0] - lut[j, 0])))
        tx = np.array([[0, -out["t"][2], out["t"][1]], [out["t"][2], 0, -out["t"][0]], [-out["t"][1], out["t"][0], 0]])
        assert loss(tx @ out["R"]) <= loss(z[f"s{seed}_E"][:3]) + 3.0           # (ii)


def test_oracle_procrustes_vs_open3d_golden():
    z = _load("o3d_procrustes.npz")
    for key in [k for k in z.files if k.endswith("_T")]:
        seed = int(key[1:-2])
        n, outl = {1000: (256, 0.2), 1001: (256, 0.5), 1010: (1024, 0.2), 1011: (1024, 0.5)}[seed]
        p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
        st, R, t, ninl = O.procrustes_solve(p["pts0"], p["pts1"], p["depth0"], p["depth1"], p["K0"], p["K1"], 0.05, 0.999, 4096, 0, seed)
        T = z[key]
        assert st == 0 and _ang(R, T[:3, :3]) < 0.5 and np.linalg.norm(t.ravel() - T[:3, 3]) < 0.05
        assert abs(ninl - int(float(z[f"s{seed}_fitness"]) * int(z[f"s{seed}_n"]))) <= 0.05 * ninl + 3


@pytest.mark.gpu
def test_hip_solvers_vs_opencv_golden():
    """the HIP solvers are bit-identical to the oracle (tests/test_gpu_*_parity.py), so one representative case per file suffices"""
    import torch
    from mapfree_reloc_amd import solver_ops as ops
    z = _load("cv_pnp.npz")
    seed, n, outl = _cases(z)[2]
    b = synth.make_batch([seed], [n], outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    o = ops.PnPBatchSolver(1000, 3.0, 0.9999, 0)(d(b["pts0"]), d(b["pts1"]), d(b["n_corr"]), d(b["depth0"]), d(b["K0"]), d(b["K1"]), d(b["pair_ids"]))
    assert int(o["status"][0]) == 0 and _ang(o["R"][0].cpu().numpy(), z[f"s{seed}_R"]) < 0.15


def test_gray_plane_vs_opencv():
    """the matcher's input plane (datasets.gray_plane: PIL decode, byte-rounded BT.601 luma, half-pixel bilinear float resize, / 255) against
    cv2.imread(GRAYSCALE) + cv2.resize(float) / 255 on the three committed JPEGs: at most one grey level apart anywhere (libjpeg's
    grayscale output vs the luma of its RGB output), identical on nearly all pixels"""
    z = _load("cv_gray.npz")
    from mapfree_reloc_amd.matchers import read_image
    for i in range(3):
        path = os.path.join(GOLD, f"gray_src_{i}.jpg")
        for (w, h) in ((120, 160), (60, 80), (200, 270)):
            got, want = read_image(path, (w, h)), z[f"g{i}_{w}x{h}"]
            assert got.shape == want.shape and got.dtype == np.float32
            d = np.abs(got.astype(np.float64) - want.astype(np.float64)) * 255.0
            assert d.max() <= 1.0 + 1e-3, (i, w, h, float(d.max()))
            assert (d < 1e-3).mean() > 0.9, (i, w, h, float((d < 1e-3).mean()))
