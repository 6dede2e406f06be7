"""Pose-solver plugins (`cfg.POSE_SOLVER`): objects with
`estimate_pose(kpts0, kpts1, data) -> (R ndarray[3,3], t ndarray[3,1] or [3], inliers int)`;
failure = NaN-filled R, t and 0 inliers (never an exception on the per-pair path).  Same names,
argument meaning and error behaviour as lib/models/matching/pose_solver.py; the arithmetic runs in
the HIP kernels behind include/mfr_hip.h (batch of one pair here; the fused pipeline batches many).

There is no CPU fallback: constructing a solver without a visible GPU raises.
"""
import numpy as np
import torch

from .. import _lib
from .. import solver_ops as ops


def backproject_3d(uv, depth, K):
    """pose_solver.py:6-17 (host restatement kept for API completeness; the device path fuses this
    into the lift kernels)"""
    uv1 = np.concatenate([uv, np.ones((uv.shape[0], 1))], axis=1)
    Ki = np.linalg.inv(np.asarray(K, dtype=np.float32))
    return np.asarray(depth, dtype=np.float32).reshape(-1, 1) * (Ki @ uv1.T).T


def _nan_pose(t_shape=(3, 1)):
    return np.full((3, 3), np.nan), np.full(t_shape, np.nan), 0


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def _pair_id(data):
    pid = data.get('pair_id', 0) if isinstance(data, dict) else 0
    return int(pid.item()) if hasattr(pid, 'item') else int(pid)


class _Base:
    def __init__(self, cfg):
        _lib.load(require_gpu=True)
        self.seed = int(cfg.RANSAC.SEED) if 'RANSAC' in cfg else 0

    @staticmethod
    def _corr(kpts0, kpts1):
        k0 = np.asarray(kpts0, dtype=np.float32).reshape(-1, 2)
        k1 = np.asarray(kpts1, dtype=np.float32).reshape(-1, 2)
        n = len(k0)
        m = max(n, 1)
        p0 = np.zeros((1, m, 2), np.float32); p1 = np.zeros((1, m, 2), np.float32)
        p0[0, :n] = k0; p1[0, :n] = k1
        return _dev(p0, torch.float32), _dev(p1, torch.float32), torch.tensor([n], dtype=torch.int32).cuda()


class PnPSolver(_Base):
    '''Estimate relative pose (metric) using Perspective-n-Point algorithm (2D-3D) correspondences
    (pose_solver.py:175-235)'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_iterations = cfg.PNP.RANSAC_ITER
        self.reprojection_inlier_threshold = cfg.PNP.REPROJECTION_INLIER_THRESHOLD
        self.confidence = cfg.PNP.CONFIDENCE
        self._solver = ops.PnPBatchSolver(self.ransac_iterations, self.reprojection_inlier_threshold,
                                          self.confidence, self.seed)

    def estimate_pose(self, pts0, pts1, data):
        if len(pts0) < 4:                                                   # :188-189
            return _nan_pose()
        p0, p1, n = self._corr(pts0, pts1)
        depth0 = data['depth0'].reshape(1, *data['depth0'].shape[-2:]).to(torch.float32).cuda()
        K0 = data['K_color0'].reshape(1, 3, 3).to(torch.float32).cuda()
        K1 = data['K_color1'].reshape(1, 3, 3).to(torch.float32).cuda()
        pid = torch.tensor([_pair_id(data)], dtype=torch.int64).cuda()
        out = self._solver(p0, p1, n, depth0, K0, K1, pid)
        if int(out["status"][0]) != ops.ST_OK:
            return _nan_pose()
        return out["R"][0].cpu().numpy(), out["t"][0].cpu().numpy().reshape(3, 1), int(out["n_inliers"][0])


class EssentialMatrixSolver(_Base):
    '''Obtain relative pose (up to scale) given a set of 2D-2D correspondences (pose_solver.py:20-61)'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_pix_threshold = cfg.EMAT_RANSAC.PIX_THRESHOLD
        self.ransac_confidence = cfg.EMAT_RANSAC.CONFIDENCE
        self._emat = ops.EssentialBatchSolver(self.ransac_pix_threshold, self.ransac_confidence, self.seed)
        self.mask = None

    def _run(self, kpts0, kpts1, data):
        p0, p1, n = self._corr(kpts0, kpts1)
        K0 = data['K_color0'].reshape(1, 3, 3).to(torch.float32).cuda()
        K1 = data['K_color1'].reshape(1, 3, 3).to(torch.float32).cuda()
        pid = torch.tensor([_pair_id(data)], dtype=torch.int64).cuda()
        out = self._emat(p0, p1, n, K0, K1, pid)
        return p0, p1, n, K0, K1, out

    def estimate_pose(self, kpts0, kpts1, data):
        if len(kpts0) < 5:                                                  # :32-33
            return _nan_pose()
        _, _, n, _, _, out = self._run(kpts0, kpts1, data)
        self.mask = out["mask"][0, :int(n[0])].cpu().numpy().reshape(-1, 1)      # :49 (cheirality-filtered, Q7)
        if int(out["status"][0]) != ops.ST_OK:
            return _nan_pose()
        return out["R"][0].cpu().numpy(), out["t"][0].cpu().numpy(), int(out["n_inliers"][0])   # t shape [3] (Q10)


class EssentialMatrixMetricSolver(EssentialMatrixSolver):
    '''E-mat decomposition + RANSAC for the translation scale from depth (pose_solver.py:115-172)'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_scale_threshold = cfg.EMAT_RANSAC.SCALE_THRESHOLD
        self._scale = ops.ScaleFromDepthBatch(self.ransac_scale_threshold)

    def estimate_pose(self, kpts0, kpts1, data):
        if len(kpts0) < 5:
            return _nan_pose()
        p0, p1, n, K0, K1, out = self._run(kpts0, kpts1, data)
        self.mask = out["mask"][0, :int(n[0])].cpu().numpy().reshape(-1, 1)
        if int(out["status"][0]) != ops.ST_OK:                              # :131-132 (inliers == 0 -> return)
            return _nan_pose()
        hw = data['depth0'].shape[-2:]
        depth0 = data['depth0'].reshape(1, *hw).to(torch.float32).cuda()
        depth1 = data['depth1'].reshape(1, *hw).to(torch.float32).cuda()
        sc = self._scale(p0, p1, out["mask"], n, depth0, depth1, K0, K1, out["R"], out["t"], out["status"])
        if int(sc["status"][0]) != ops.ST_OK:                               # :145-149
            return _nan_pose()
        return out["R"][0].cpu().numpy(), sc["t_metric"][0].cpu().numpy().reshape(3, 1), int(sc["n_inliers"][0])


class ProcrustesSolver(_Base):
    '''Estimate relative pose (metric) using 3D-3D correspondences (pose_solver.py:238-320).
    PROCRUSTES.REFINE (full-cloud ICP, :291-315) is not built: requesting it raises.'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_max_corr_distance = cfg.PROCRUSTES.MAX_CORR_DIST
        self.refine = cfg.PROCRUSTES.REFINE
        if self.refine:
            raise NotImplementedError("PROCRUSTES.REFINE (Open3D ICP, pose_solver.py:291-315) is not built")
        self._solver = ops.ProcrustesBatchSolver(self.ransac_max_corr_distance, 0.999, self.seed)

    def estimate_pose(self, pts0, pts1, data):
        if len(pts0) < 3:                                                   # :252-253
            return _nan_pose()
        p0, p1, n = self._corr(pts0, pts1)
        hw = data['depth0'].shape[-2:]
        depth0 = data['depth0'].reshape(1, *hw).to(torch.float32).cuda()
        depth1 = data['depth1'].reshape(1, *hw).to(torch.float32).cuda()
        K0 = data['K_color0'].reshape(1, 3, 3).to(torch.float32).cuda()
        K1 = data['K_color1'].reshape(1, 3, 3).to(torch.float32).cuda()
        pid = torch.tensor([_pair_id(data)], dtype=torch.int64).cuda()
        out = self._solver(p0, p1, n, depth0, depth1, K0, K1, pid)
        if int(out["status"][0]) != ops.ST_OK:
            return _nan_pose()
        return out["R"][0].cpu().numpy(), out["t"][0].cpu().numpy().reshape(3, 1), int(out["n_inliers"][0])
