"""Generate tests/golden/ref_k64.npz: the FLOAT64-intrinsics flow of the real Map-free loader, by EXECUTING THE
REFERENCE'S OWN PYTHON (build container only; the GPU box never runs this).

Why: `correct_intrinsic_scale` (lib/datasets/utils.py:117-130) multiplies a float64 `np.eye(3)` into the float32 K that
`MapFreeScene.read_intrinsics` parses (lib/datasets/mapfree.py:40-55), and the Map-free configs always resize
(config/mapfree.yaml:7-8) -- so `data['K_color*']` reaches `backproject_3d` (pose_solver.py:15-16), the K-normalisation
(:39-43) and PnP (:204-213) as FLOAT64.  The fixtures written by gen_golden.py use float32 K (the resize=None flow).

What runs for real here (cv2 / open3d / pytorch_lightning / transforms3d stubbed at import, none of them on these lines):
  * MapFreeScene.read_intrinsics on an intrinsics.txt written below  -> float64 K, stored in the fixture;
  * backproject_3d with that K;
  * EssentialMatrixSolver.estimate_pose's own lines up to cv.findEssentialMat: the normalised keypoints and the RANSAC
    threshold it would hand to OpenCV are CAPTURED by the stub (for the float64 K and, to pin that flow too, for the
    float32 K of a resize=None dataset);
  * EssentialMatrixMetricSolver.estimate_pose (depth gather, validity, back-projection, rotation, per-point scales,
    exhaustive scale RANSAC) with cv2 replaced by a replay stub;
  * PnPSolver.estimate_pose up to cv.solvePnPRansac (lifted points, observations and the K it passes are captured).

Usage: python oracle/gen_k64_golden.py
"""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import NpT, npt, _Cfg, rand_rot  # noqa: E402  (helpers only; nothing numerical)


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


def _import_reference():
    cv = _Stub("cv2")
    sys.modules["cv2"] = cv
    sys.modules["open3d"] = _Stub("open3d")
    for name in ("pytorch_lightning", "transforms3d", "transforms3d.quaternions", "tqdm"):
        sys.modules[name] = _Stub(name)
    sys.modules["pytorch_lightning"]._logger = _Stub("logger")
    sys.path.insert(0, REF)
    import importlib
    ps = importlib.import_module("lib.models.matching.pose_solver")
    mf = importlib.import_module("lib.datasets.mapfree")
    return cv, ps, mf


def read_K(mf, rng, native_wh, resize):
    """two frames' intrinsics through the reference's own parser + rescale -> (K0, K1), float64 when resize is given"""
    Wn, Hn = native_wh
    with tempfile.TemporaryDirectory() as td:
        lines = ["# name fx fy cx cy W H"]
        for s in (0, 1):
            f = rng.uniform(0.9, 1.1) * 590.0 * Wn / 540.0
            lines.append(f"seq{s}/frame_00000.jpg {f:.6f} {f * rng.uniform(0.98, 1.02):.6f} "
                         f"{Wn / 2 + rng.uniform(-8, 8):.6f} {Hn / 2 + rng.uniform(-8, 8):.6f} {Wn} {Hn}")
        (Path(td) / "intrinsics.txt").write_text("\n".join(lines) + "\n")
        Ks = mf.MapFreeScene.read_intrinsics(Path(td), resize)
    return Ks["seq0/frame_00000.jpg"], Ks["seq1/frame_00000.jpg"]


def scene(rng, H, W, n, K0, K1, zero_frac):
    depth0 = (np.round(rng.uniform(0.5, 8.0, size=(H, W)) * 1000) / 1000).astype(np.float32)
    depth1 = (np.round(rng.uniform(0.5, 8.0, size=(H, W)) * 1000) / 1000).astype(np.float32)
    depth0[rng.uniform(size=(H, W)) < zero_frac] = 0.0
    depth1[rng.uniform(size=(H, W)) < zero_frac] = 0.0
    pts0 = np.stack([rng.uniform(0, W - 1e-3, n), rng.uniform(0, H - 1e-3, n)], 1).astype(np.float32)
    pts1 = np.stack([rng.uniform(0, W - 1e-3, n), rng.uniform(0, H - 1e-3, n)], 1).astype(np.float32)
    # the collated batch holds K as a [1,3,3] tensor of the loader's dtype; NpT = ndarray answering .numpy() (see gen_golden.py)
    data = {"K_color0": npt(K0[None]), "K_color1": npt(K1[None]), "depth0": npt(depth0[None]), "depth1": npt(depth1[None])}
    return depth0, depth1, pts0, pts1, data


def main():
    cv, ps, mf = _import_reference()
    rng = np.random.default_rng(20260926)
    H, W = 96, 72          # depth maps kept small (fixture size); K is the real 540x720 camera, pixels stay inside the map
    cases = {}
    # native sizes: the dataset's own 540x720 (scale 1: K keeps its values but BECOMES float64), 1080x1440, and an odd size
    RS = (540, 720)        # config/mapfree.yaml:7-8
    flows = [((540, 720), RS), ((1080, 1440), RS), ((720, 960), RS), ((1440, 1920), RS),
             ((540, 720), None)]          # the last one: resize=None -> float32 K (quirk Q5 flow), normalisation pinned too
    nc = 0
    for native, resize in flows:
        for rep in range(3):
            K0, K1 = read_K(mf, rng, native, resize)
            assert K0.dtype == (np.float64 if resize is not None else np.float32)
            n = int(rng.integers(40, 500)) if rep < 2 else 7
            depth0, depth1, pts0, pts1, data = scene(rng, H, W, n, K0, K1, zero_frac=0.15)
            p = f"c{nc}_"
            cases[p + "K0"] = np.asarray(K0); cases[p + "K1"] = np.asarray(K1)
            cases[p + "depth0"] = (depth0 * 1000).round().astype(np.uint16); cases[p + "depth1"] = (depth1 * 1000).round().astype(np.uint16)
            cases[p + "pts0"] = pts0; cases[p + "pts1"] = pts1
            # (1) backproject_3d (pose_solver.py:6-17) at the truncated pixels
            uv = np.int32(pts0)
            d = depth0[uv[:, 1], uv[:, 0]]
            xyz = ps.backproject_3d(uv, d, data["K_color0"].squeeze(0))
            assert xyz.dtype == np.float64
            cases[p + "bp_xyz"] = np.asarray(xyz)
            # (2) E-mat normalisation + threshold, then the metric-scale leg with replayed (E, mask, R, t)
            R = rand_rot(rng); t = rng.normal(size=3); t /= np.linalg.norm(t)
            mask_in = (rng.uniform(size=n) < 0.7).astype(np.uint8).reshape(-1, 1)
            cap = {}

            def findEssentialMat(k0, k1, Kc, threshold=None, prob=None, method=None):
                cap["k0"] = np.asarray(k0).copy(); cap["k1"] = np.asarray(k1).copy(); cap["thr"] = threshold
                return np.eye(3), mask_in.copy()

            def recoverPose(_E, k0, k1, Kc, dist, mask=None):
                return int(mask.sum()), R.copy(), t.reshape(3, 1).copy(), mask
            cv.findEssentialMat = findEssentialMat; cv.recoverPose = recoverPose; cv.USAC_MAGSAC = 38
            Rr, tr, inl = ps.EssentialMatrixMetricSolver(_Cfg()).estimate_pose(pts0.copy(), pts1.copy(), data)
            want = np.float64 if resize is not None else np.float32
            assert cap["k0"].dtype == want, cap["k0"].dtype
            cases[p + "k0n"] = cap["k0"].astype(np.float64); cases[p + "k1n"] = cap["k1"].astype(np.float64)
            cases[p + "thr"] = np.float64(cap["thr"])
            cases[p + "mask"] = mask_in.ravel(); cases[p + "R_in"] = R; cases[p + "t_in"] = t
            cases[p + "R_out"] = np.asarray(Rr, np.float64); cases[p + "t_out"] = np.asarray(tr, np.float64).reshape(-1)
            cases[p + "inliers"] = np.int64(inl)
            # (3) PnP up to cv.solvePnPRansac
            cap2 = {}

            def solvePnPRansac(xyz, p1, K, dist, iterationsCount=None, reprojectionError=None, confidence=None, flags=None):
                cap2["xyz"] = np.asarray(xyz, np.float64).copy(); cap2["obs"] = np.asarray(p1).copy(); cap2["K"] = np.asarray(K).copy()
                return False, None, None, None
            cv.solvePnPRansac = solvePnPRansac; cv.SOLVEPNP_P3P = 2
            ps.PnPSolver(_Cfg()).estimate_pose(pts0.copy(), pts1.copy(), data)
            assert cap2["K"].dtype == want
            cases[p + "pnp_xyz"] = cap2["xyz"]; cases[p + "pnp_obs"] = cap2["obs"].astype(np.float64)
            cases[p + "pnp_K"] = cap2["K"].astype(np.float64)
            nc += 1
    cases["n_cases"] = np.int64(nc)
    np.savez_compressed(os.path.join(OUT, "ref_k64.npz"), **cases)
    print("wrote ref_k64.npz:", nc, "cases;", sum(1 for c in range(nc) if cases[f'c{c}_K0'].dtype == np.float64), "with float64 K")


if __name__ == "__main__":
    main()
