"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference; the GPU box never
runs this).  The reference modules import cv2 / open3d, which are not
installed, so those two modules are replaced by stubs *before* import:

  * every reference-owned line (back-projection, int-truncation, depth gather,
    validity rules, rotation of xyz0, per-point scale, the exhaustive scale
    RANSAC loop, NaN stripping of the npz wire format, stack_pts) runs for real;
  * the cv2 entry points (findEssentialMat / recoverPose / solvePnPRansac) are
    stubs that replay fixed (E, mask, R, t) values or capture their inputs --
    their arithmetic is NOT pinned by these fixtures (parity unpinned vs OpenCV,
    see DESIGN.md).

Usage: python oracle/gen_golden.py   (writes tests/golden/ref_*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return 0


def _install_stubs():
    cv = _Stub("cv2")
    sys.modules["cv2"] = cv
    o3d = _Stub("open3d")
    sys.modules["open3d"] = o3d
    return cv


def _import_reference():
    cv = _install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "etc", "feature_matching_baselines"))
    import importlib
    ps = importlib.import_module("lib.models.matching.pose_solver")
    fm = importlib.import_module("lib.models.matching.feature_matching")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_fmb_utils", os.path.join(REF, "etc", "feature_matching_baselines", "utils.py"))
    ut = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ut)
    return cv, ps, fm, ut


class NpT(np.ndarray):
    """ndarray that also answers .numpy(): emulates how torch tensors flowed through
    the reference under its pinned numpy 1.24 / torch 2.0 (np.linalg.inv(tensor) ->
    ndarray f32; quirk Q5).  Under this container's numpy 2.2 / torch 2.10 the
    reference's own `ndarray * Tensor` raises TypeError, so tensors cannot be used."""
    def numpy(self):
        return np.asarray(self)


def npt(a):
    return np.asarray(a).view(NpT)


class _Cfg:
    """attribute bag standing in for the yacs node (only the keys the solvers read)"""
    class _NS:
        pass

    def __init__(self):
        self.EMAT_RANSAC = self._NS()
        self.EMAT_RANSAC.PIX_THRESHOLD = 2.0
        self.EMAT_RANSAC.SCALE_THRESHOLD = 0.1
        self.EMAT_RANSAC.CONFIDENCE = 0.9999
        self.PNP = self._NS()
        self.PNP.RANSAC_ITER = 1000
        self.PNP.REPROJECTION_INLIER_THRESHOLD = 3
        self.PNP.CONFIDENCE = 0.9999
        self.DEBUG = False
        self.MATCHES_FILE_PATH = ""
        self.DATASET = self._NS()
        self.DATASET.PAIRS_TXT = self._NS()
        self.DATASET.PAIRS_TXT.TEST = ""


def rand_rot(rng, maxdeg=30.0):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = np.deg2rad(rng.uniform(0, maxdeg))
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def make_scene(rng, H, W, n, zero_frac=0.1):
    f = rng.uniform(0.9, 1.1) * 590.0 * W / 540.0
    K0 = np.array([[f, 0, W / 2 - 0.5], [0, f * rng.uniform(0.98, 1.02), H / 2 - 0.5], [0, 0, 1]], dtype=np.float32)
    K1 = np.array([[f * rng.uniform(0.95, 1.05), 0, W / 2 - 0.3], [0, f, H / 2 - 0.8], [0, 0, 1]], dtype=np.float32)
    depth0 = (np.round(rng.uniform(0.5, 8.0, size=(H, W)) * 1000) / 1000).astype(np.float32)
    depth1 = (np.round(rng.uniform(0.5, 8.0, size=(H, W)) * 1000) / 1000).astype(np.float32)
    depth0[rng.uniform(size=(H, W)) < zero_frac] = 0.0
    depth1[rng.uniform(size=(H, W)) < zero_frac] = 0.0
    pts0 = np.stack([rng.uniform(0, W - 1e-3, n), rng.uniform(0, H - 1e-3, n)], 1).astype(np.float32)
    pts1 = np.stack([rng.uniform(0, W - 1e-3, n), rng.uniform(0, H - 1e-3, n)], 1).astype(np.float32)
    data = {
        "K_color0": npt(K0[None]), "K_color1": npt(K1[None]),
        "depth0": npt(depth0[None]), "depth1": npt(depth1[None]),
    }
    return K0, K1, depth0, depth1, pts0, pts1, data


def gen_backproject(ps, rng):
    cases = {}
    for c in range(8):
        n = int(rng.integers(1, 200))
        K = np.array([[rng.uniform(300, 900), 0, rng.uniform(200, 400)],
                      [0, rng.uniform(300, 900), rng.uniform(200, 500)], [0, 0, 1]], dtype=np.float32)
        uv = np.stack([rng.integers(0, 540, n), rng.integers(0, 720, n)], 1).astype(np.int32)
        d = rng.uniform(0.1, 20, n).astype(np.float32)
        xyz = ps.backproject_3d(uv, d, K)      # f32 K -> f32 inverse, f64 product (quirk Q5)
        assert xyz.dtype == np.float64
        cases[f"c{c}_K"] = K; cases[f"c{c}_uv"] = uv; cases[f"c{c}_d"] = d; cases[f"c{c}_xyz"] = np.asarray(xyz)
    cases["n_cases"] = np.int64(8)
    np.savez_compressed(os.path.join(OUT, "ref_backproject.npz"), **cases)


def gen_emat_metric(cv, ps, rng):
    """EssentialMatrixMetricSolver.estimate_pose with cv2 replaced by a replay stub."""
    cases = {}
    nc = 0
    for c in range(10):
        H, W = 48, 36
        n = int(rng.integers(5, 400)) if c < 8 else (6 if c == 8 else 40)
        K0, K1, depth0, depth1, pts0, pts1, data = make_scene(rng, H, W, n, zero_frac=0.15 if c != 9 else 1.0)
        R = rand_rot(rng); t = rng.normal(size=3); t /= np.linalg.norm(t)
        mask_in = (rng.uniform(size=n) < 0.7).astype(np.uint8).reshape(-1, 1)
        if c == 7:   # make the points cluster so the scale consensus is non-trivial
            depth0[:] = 3.0; depth1[:] = 3.2
            data["depth0"] = npt(depth0[None]); data["depth1"] = npt(depth1[None])
        E = np.eye(3)

        def findEssentialMat(k0, k1, Kc, threshold=None, prob=None, method=None):
            return E.copy(), mask_in.copy()

        def recoverPose(_E, k0, k1, Kc, dist, mask=None):
            return int(mask.sum()), R.copy(), t.reshape(3, 1).copy(), mask
        cv.findEssentialMat = findEssentialMat
        cv.recoverPose = recoverPose
        cv.USAC_MAGSAC = 38
        solver = ps.EssentialMatrixMetricSolver(_Cfg())
        Rr, tr, inl = solver.estimate_pose(pts0.copy(), pts1.copy(), data)
        p = f"c{nc}_"
        cases[p + "K0"] = K0; cases[p + "K1"] = K1; cases[p + "depth0"] = depth0; cases[p + "depth1"] = depth1
        cases[p + "pts0"] = pts0; cases[p + "pts1"] = pts1; cases[p + "mask"] = mask_in.ravel()
        cases[p + "R_in"] = R; cases[p + "t_in"] = t
        cases[p + "R_out"] = np.asarray(Rr, dtype=np.float64); cases[p + "t_out"] = np.asarray(tr, dtype=np.float64).reshape(-1)
        cases[p + "inliers"] = np.int64(inl)
        nc += 1
    cases["n_cases"] = np.int64(nc)
    np.savez_compressed(os.path.join(OUT, "ref_emat_metric.npz"), **cases)


def gen_pnp_lift(cv, ps, rng):
    """PnPSolver.estimate_pose up to the cv.solvePnPRansac call (inputs captured)."""
    cases = {}
    nc = 0
    for c in range(8):
        H, W = 48, 36
        n = int(rng.integers(4, 300)) if c < 6 else (3 if c == 6 else 30)
        K0, K1, depth0, depth1, pts0, pts1, data = make_scene(rng, H, W, n, zero_frac=0.2 if c != 7 else 1.0)
        captured = {}

        def solvePnPRansac(xyz, p1, K, dist, iterationsCount=None, reprojectionError=None, confidence=None, flags=None):
            captured["xyz"] = np.asarray(xyz, dtype=np.float64).copy()
            captured["pts1"] = np.asarray(p1).copy()
            captured["K"] = np.asarray(K).copy()
            captured["args"] = (iterationsCount, reprojectionError, confidence)
            return False, None, None, None
        cv.solvePnPRansac = solvePnPRansac
        cv.SOLVEPNP_P3P = 2
        solver = ps.PnPSolver(_Cfg())
        Rr, tr, inl = solver.estimate_pose(pts0.copy(), pts1.copy(), data)
        assert np.isnan(Rr).all() and inl == 0
        p = f"c{nc}_"
        cases[p + "K0"] = K0; cases[p + "K1"] = K1; cases[p + "depth0"] = depth0
        cases[p + "pts0"] = pts0; cases[p + "pts1"] = pts1
        cases[p + "called"] = np.int64(1 if captured else 0)
        if captured:
            cases[p + "xyz"] = captured["xyz"]; cases[p + "obs"] = captured["pts1"].astype(np.float64)
            assert captured["args"] == (1000, 3, 0.9999)
        nc += 1
    cases["n_cases"] = np.int64(nc)
    np.savez_compressed(os.path.join(OUT, "ref_pnp_lift.npz"), **cases)


def gen_wire_format(fm, ut, rng, tmpdir="/tmp/mfr_golden_tmp"):
    """stack_pts (utils.py:59-69) + np.savez_compressed (compute.py:84-85) +
    PrecomputedMatching.get_correspondences (feature_matching.py:28-50)."""
    os.makedirs(tmpdir, exist_ok=True)
    pts_list = []
    for i in range(6):
        n = [17, 0, 5, 1, 33, 8][i]
        if n == 0:
            pts_list.append(np.full((1, 4), np.nan))            # matchers.py:59,120
        else:
            pts_list.append(rng.uniform(0, 540, size=(n, 4)).astype(np.float32))
    stack = ut.stack_pts(pts_list)
    path = os.path.join(tmpdir, "correspondences_SG.npz")
    np.savez_compressed(path, correspondences=stack)
    cfg = _Cfg(); cfg.MATCHES_FILE_PATH = "{scene_root}/correspondences_SG.npz"
    pm = fm.PrecomputedMatching(cfg)
    cases = {"stack": stack, "n_pairs": np.int64(len(pts_list))}
    for i in range(len(pts_list)):
        cases[f"in{i}"] = np.asarray(pts_list[i], dtype=np.float64)
        data = {"scene_id": ["s00001"], "scene_root": [tmpdir], "pair_id": torch.tensor([i])}
        p1, p2 = pm.get_correspondences(data)
        cases[f"p1_{i}"] = np.asarray(p1); cases[f"p2_{i}"] = np.asarray(p2)
    np.savez_compressed(os.path.join(OUT, "ref_wire_format.npz"), **cases)


def gen_metrics(rng):
    """the reference's benchmark metric code (benchmark/metrics.py, reprojection.py, utils.py,
    mapfree.py:aggregate_results) executed for real; only the `transforms3d` package (absent offline)
    is replaced by a self-contained quaternion module written here (Hamilton product, w-first)."""
    q3 = types.ModuleType("transforms3d.quaternions")

    def qmult(a, b):
        w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
        return np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2,
                         w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2])

    def qconj(q):
        return np.array([q[0], -q[1], -q[2], -q[3]])
    q3.qmult = qmult
    q3.qinverse = lambda q: qconj(q) / np.dot(q, q)
    q3.rotate_vector = lambda v, q: qmult(q, qmult(np.r_[0.0, v], qconj(q)))[1:]

    def quat2mat(q):
        w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
        return np.array([[1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y)], [2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x)],
                         [2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y)]])
    q3.quat2mat = quat2mat
    t3 = types.ModuleType("transforms3d"); t3.quaternions = q3
    sys.modules["transforms3d"] = t3; sys.modules["transforms3d.quaternions"] = q3
    import importlib
    # benchmark/mapfree.py:14 imports the yacs-based config only for its CLI defaults: stub it
    yc = types.ModuleType("yacs.config")

    class _CN(dict):
        def __getattr__(self, k):
            return self.get(k)

        def __setattr__(self, k, v):
            self[k] = v
    yc.CfgNode = _CN
    sys.modules["yacs"] = types.ModuleType("yacs"); sys.modules["yacs.config"] = yc
    metrics = importlib.import_module("benchmark.metrics")
    mapfree = importlib.import_module("benchmark.mapfree")
    butils = importlib.import_module("benchmark.utils")
    K = np.array([[590.0, 0, 269.5], [0, 590.0, 359.5], [0, 0, 1]], dtype=np.float32)
    W, H = 540, 720
    cases = {"K": K, "W": np.int64(W), "H": np.int64(H)}
    mm = metrics.MetricManager()
    all_results = {}
    fi = 0
    from collections import defaultdict
    for s in range(3):
        res = defaultdict(list)
        for f in range(12):
            q_gt = rng.normal(size=4); q_gt /= np.linalg.norm(q_gt)
            t_gt = rng.normal(size=3) * 2
            ang = np.deg2rad(rng.uniform(0, 12)) * (0.2 if f % 3 else 1.0)
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            dq = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax]
            q_est = qmult(q_gt, dq) * rng.choice([-1.0, 1.0]) * rng.uniform(0.5, 2.0)
            t_est = t_gt + rng.normal(size=3) * (0.05 if f % 2 else 0.4)
            conf = float(rng.integers(0, 40)) if f % 4 else 17.0
            inp = metrics.Inputs(q_gt=q_gt, t_gt=t_gt, q_est=q_est, t_est=t_est, confidence=conf, K=K, W=W, H=H)
            before = {k: len(v) for k, v in res.items()}
            mm(inp, res)
            p = f"f{fi}_"
            cases[p + "q_gt"] = q_gt; cases[p + "t_gt"] = t_gt; cases[p + "q_est"] = q_est; cases[p + "t_est"] = t_est
            cases[p + "conf"] = np.float64(conf); cases[p + "scene"] = np.int64(s)
            for m in ("trans_err", "rot_err", "reproj_err"):
                cases[p + m] = np.float64(res[m][-1])
            fi += 1
        all_results[f"s{s}"] = res
    cases["n_frames"] = np.int64(fi)
    agg = mapfree.aggregate_results(all_results, all_failures=5)
    for i, (k, v) in enumerate(agg.items()):
        cases[f"agg{i}_name"] = np.array(k); cases[f"agg{i}_val"] = np.float64(v)
    cases["n_agg"] = np.int64(len(agg))
    # world2cam -> cam2world conversion used when reading pose files (benchmark/utils.py:12-15)
    q = rng.normal(size=4); t = rng.normal(size=3)
    qi, ti = butils.convert_world2cam_to_cam2world(q, t)
    cases["w2c_q"] = q; cases["w2c_t"] = t; cases["c2w_q"] = qi; cases["c2w_t"] = ti
    np.savez_compressed(os.path.join(OUT, "ref_metrics.npz"), **cases)


def _sift_like(rng, n):
    """integer-valued 128-d descriptors with OpenCV SIFT's statistics (u8-saturated, |d|_2 ~ 512)"""
    d = rng.gamma(0.6, 1.0, (n, 128))
    d = np.minimum(d / np.linalg.norm(d, axis=1, keepdims=True), 0.2)
    d = np.clip(np.rint(512.0 * d / np.linalg.norm(d, axis=1, keepdims=True)), 0, 255)
    return d.astype(np.float32)


def gen_sift(cv, fm, rng):
    """root_sift (feature_matching.py:68-74) and the whole get_correspondences loop (:75-118) of the
    reference, executed with a stub cv2 whose SIFT returns prepared (keypoints, descriptors) and whose
    FlannBasedMatcher is an exact float64 brute-force 2-NN (FLANN itself is unavailable offline)."""
    cases = {}
    # (1) root_sift on SIFT-like integer descriptors, on general floats and on an all-zero row
    a = _sift_like(rng, 257)
    b = (rng.random((64, 128)) * 37.0).astype(np.float32)
    b[5] = 0.0
    for name, d in (("rs_int", a), ("rs_float", b)):
        cases[name + "_in"] = d
        cases[name + "_out"] = fm.SIFTMatching.root_sift(None, d.copy())

    # (2) get_correspondences with stubbed detection / 2-NN
    class KP:
        def __init__(self, pt):
            self.pt = (float(pt[0]), float(pt[1]))

    class DM:
        def __init__(self, q, t, dist):
            self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(np.float32(dist))

    class Flann:
        def __init__(self, *a, **k):
            pass

        def knnMatch(self, d0, d1, k=2):
            D = ((d0[:, None, :].astype(np.float64) - d1[None].astype(np.float64)) ** 2).sum(-1)
            order = np.argsort(D, axis=1, kind="stable")[:, :2]
            return [(DM(i, order[i, 0], np.sqrt(D[i, order[i, 0]])), DM(i, order[i, 1], np.sqrt(D[i, order[i, 1]])))
                    for i in range(len(d0))]

    for ci, (n0, n1, n_true) in enumerate(((300, 280, 150), (64, 97, 20), (2048, 2048, 700))):
        d0 = _sift_like(rng, n0)
        d1 = _sift_like(rng, n1)
        perm = rng.permutation(n1)[:n_true]
        src = rng.permutation(n0)[:n_true]
        noisy = d0[src] + rng.normal(0, rng.uniform(2.0, 60.0, (n_true, 1)), (n_true, 128))
        d1[perm] = np.clip(np.rint(noisy), 0, 255).astype(np.float32)
        kp0 = (rng.random((n0, 2)) * [720, 540]).astype(np.float32)
        kp1 = (rng.random((n1, 2)) * [720, 540]).astype(np.float32)
        feats = [([KP(p) for p in kp0], d0.copy()), ([KP(p) for p in kp1], d1.copy())]

        class Sift:
            def __init__(self):
                self.i = 0

            def detectAndCompute(self, img, mask):
                r = feats[self.i]
                self.i += 1
                return r

        m = fm.SIFTMatching.__new__(fm.SIFTMatching)
        m.ratio_threshold = 0.8
        m.sift = Sift()
        m.debug = False
        cv.FlannBasedMatcher = Flann
        cv.cvtColor = lambda im, code: im[..., 0]
        img = torch.zeros(1, 3, 8, 8)
        pts1, pts2 = m.get_correspondences({"image0": img, "image1": img})
        cases[f"gc{ci}_des0"] = d0.astype(np.uint8); cases[f"gc{ci}_des1"] = d1.astype(np.uint8)   # integer-valued
        cases[f"gc{ci}_kp0"] = kp0; cases[f"gc{ci}_kp1"] = kp1
        cases[f"gc{ci}_pts1"] = pts1; cases[f"gc{ci}_pts2"] = pts2
    np.savez_compressed(os.path.join(OUT, "ref_sift_ratio.npz"), **cases)


def main():
    os.makedirs(OUT, exist_ok=True)
    cv, ps, fm, ut = _import_reference()
    rng = np.random.default_rng(20240807)
    gen_backproject(ps, rng)
    gen_emat_metric(cv, ps, rng)
    gen_pnp_lift(cv, ps, rng)
    gen_wire_format(fm, ut, rng)
    gen_metrics(rng)
    gen_sift(cv, fm, np.random.default_rng(20240808))
    print("golden fixtures written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
