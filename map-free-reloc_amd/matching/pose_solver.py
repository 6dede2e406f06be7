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
    Ki = np.linalg.inv(np.asarray(K))            # in K's own dtype (float64 from the Map-free loader, float32 otherwise)
    return np.asarray(depth, dtype=np.float32).reshape(-1, 1) * (Ki @ uv1.T).T


def _nan_pose(t_shape=(3, 1)):
    return np.full((3, 3), np.nan), np.full(t_shape, np.nan), 0


def _pair_id(data):
    pid = data.get('pair_id', 0) if isinstance(data, dict) else 0
    return int(pid.item()) if hasattr(pid, 'item') else int(pid)


class _PairStage:
    """Host->device hand-over of ONE pair for the per-pair plugin API (batch 1): everything the solver needs
    -- count and RANSAC stream id, both intrinsics IN THE DTYPE `data` HOLDS THEM (float64 from the Map-free loader,
    lib/datasets/utils.py:117-130; float32 from resize=None datasets), the correspondences -- is packed into one pinned host
    buffer and crosses PCIe as ONE asynchronous copy of the used prefix into a persistent device buffer (the reference's flow
    hands numpy arrays to OpenCV; the first version of this class issued 6-8 small synchronous copies per pair).  The device
    tensors handed to the kernels are views into that buffer; results come back as one packed D2H copy (`fetch`).
    Pack layout in 4-byte words: meta [0,8) = n (i32) | pad | pair id (i64 in words 2-3) ; K0 | K1 [8,48) (2 x 9 float32 or
    2 x 9 float64) ; pts0 [48, 48+2m) ; pts1 [48+2m, 48+4m) ; then, at fixed offsets, the two depth maps."""
    O_META, O_K, O_P = 0, 8, 48

    def __init__(self):
        self.hw = None
        self.cap = 8192                              # correspondences per pair; grows on demand (LoFTR yields <= 6120)
        self.host = self.dev = None

    def _alloc(self, H, W, cap):
        self.hw, self.cap = (H, W), cap
        self.o_d0 = self.O_P + 4 * cap
        self.o_d1 = self.o_d0 + H * W
        n = self.o_d1 + H * W
        self.host = torch.empty(self.o_d0, dtype=torch.float32, pin_memory=True)
        self.dev = torch.empty(n, dtype=torch.float32, device='cuda')
        self.host_i32 = self.host.view(torch.int32)
        self.host_i64 = self.host[self.O_META + 2:self.O_META + 4].view(torch.int64)
        self.host_k64 = self.host[self.O_K:self.O_K + 36].view(torch.float64)

    def put(self, kpts0, kpts1, data, need_depth0=True, need_depth1=False):
        """-> dict of device views (pts0/pts1 [1,m,2], n [1] i32, depth0/depth1 [1,H,W], K0/K1 [1,3,3] f32 or f64, pid [1] i64)"""
        k0 = np.asarray(kpts0, dtype=np.float32).reshape(-1, 2)
        k1 = np.asarray(kpts1, dtype=np.float32).reshape(-1, 2)
        n = len(k0)
        # the depth-free solver (EssentialMatrixSolver) must not touch data['depth*'] (an uncollated empty tensor when the
        # dataset has no depth, lib/datasets/mapfree.py:233-234)
        H, W = data['depth0'].shape[-2:] if (need_depth0 or need_depth1) else (self.hw or (0, 0))
        cap = self.cap
        while n > cap:
            cap *= 2                                 # every correspondence is used, as in the reference (no truncation)
        if self.hw != (H, W) or cap != self.cap or self.host is None:
            self._alloc(H, W, cap)
        m = max(n, 1)
        h = self.host
        o_p1 = self.O_P + 2 * m
        h[self.O_P:self.O_P + 2 * n] = torch.from_numpy(k0.reshape(-1))
        h[o_p1:o_p1 + 2 * n] = torch.from_numpy(k1.reshape(-1))
        K0, K1 = data['K_color0'], data['K_color1']
        k64 = (K0.dtype == torch.float64) or (K1.dtype == torch.float64)
        if k64:
            self.host_k64[:9] = K0.reshape(9); self.host_k64[9:18] = K1.reshape(9)
        else:
            h[self.O_K:self.O_K + 9] = K0.reshape(9); h[self.O_K + 9:self.O_K + 18] = K1.reshape(9)
        self.host_i32[self.O_META] = n
        self.host_i64[0] = _pair_id(data)
        # the small operands cross in ONE async copy from the pinned pack (only the used prefix); the depth maps (1.5 MB each)
        # go straight from the loader's pageable tensors into their slots of the persistent device buffer (CPU stores into
        # pinned, uncached host memory were measured slower than the driver's own staging for MB-sized pieces)
        d = self.dev
        used = self.O_P + 4 * m
        d[:used].copy_(h[:used], non_blocking=True)
        if need_depth0:
            d[self.o_d0:self.o_d0 + H * W].copy_(data['depth0'].reshape(-1), non_blocking=True)
        if need_depth1:
            d[self.o_d1:self.o_d1 + H * W].copy_(data['depth1'].reshape(-1), non_blocking=True)
        di32 = d.view(torch.int32)
        if k64:
            kk = d[self.O_K:self.O_K + 36].view(torch.float64)
            Kd0, Kd1 = kk[:9].view(1, 3, 3), kk[9:18].view(1, 3, 3)
        else:
            Kd0, Kd1 = d[self.O_K:self.O_K + 9].view(1, 3, 3), d[self.O_K + 9:self.O_K + 18].view(1, 3, 3)
        return dict(pts0=d[self.O_P:self.O_P + 2 * m].view(1, m, 2), pts1=d[o_p1:o_p1 + 2 * m].view(1, m, 2),
                    n=di32[self.O_META:self.O_META + 1], K0=Kd0, K1=Kd1,
                    pid=d[self.O_META + 2:self.O_META + 4].view(torch.int64),
                    depth0=d[self.o_d0:self.o_d0 + H * W].view(1, H, W), depth1=d[self.o_d1:self.o_d1 + H * W].view(1, H, W), n_host=n)

    @staticmethod
    def fetch(*tensors):
        """several small device results -> host float64 arrays with ONE D2H copy (and one sync)"""
        flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors]).cpu().numpy()
        out, o = [], 0
        for t in tensors:
            k = t.numel()
            out.append(flat[o:o + k]); o += k
        return out


class _Base:
    def __init__(self, cfg):
        _lib.load(require_gpu=True)
        self.seed = int(cfg.RANSAC.SEED) if 'RANSAC' in cfg else 0
        self._stage = _PairStage()


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
        d = self._stage.put(pts0, pts1, data, need_depth1=False)
        out = self._solver(d["pts0"], d["pts1"], d["n"], d["depth0"], d["K0"], d["K1"], d["pid"])
        st, R, t, ninl = self._stage.fetch(out["status"], out["R"], out["t"], out["n_inliers"])
        if int(st[0]) != ops.ST_OK:
            return _nan_pose()
        return R.reshape(3, 3), t.reshape(3, 1), int(ninl[0])


class EssentialMatrixSolver(_Base):
    '''Obtain relative pose (up to scale) given a set of 2D-2D correspondences (pose_solver.py:20-61)'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_pix_threshold = cfg.EMAT_RANSAC.PIX_THRESHOLD
        self.ransac_confidence = cfg.EMAT_RANSAC.CONFIDENCE
        self._emat = ops.EssentialBatchSolver(self.ransac_pix_threshold, self.ransac_confidence, self.seed,
                                               score=cfg.HIP.EMAT_SCORE, max_thr_ratio=cfg.HIP.MAGSAC_MAX_THR_RATIO)
        self.mask = None

    def _run(self, kpts0, kpts1, data, need_depth):
        d = self._stage.put(kpts0, kpts1, data, need_depth0=need_depth, need_depth1=need_depth)
        out = self._emat(d["pts0"], d["pts1"], d["n"], d["K0"], d["K1"], d["pid"])
        return d, out

    def estimate_pose(self, kpts0, kpts1, data):
        if len(kpts0) < 5:                                                  # :32-33
            return _nan_pose()
        d, out = self._run(kpts0, kpts1, data, need_depth=False)
        st, R, t, ninl, mask = self._stage.fetch(out["status"], out["R"], out["t"], out["n_inliers"], out["mask"][0, :d["n_host"]])
        self.mask = mask.astype(np.uint8).reshape(-1, 1)                     # :49 (cheirality-filtered, Q7)
        if int(st[0]) != ops.ST_OK:
            return _nan_pose()
        return R.reshape(3, 3), t.reshape(3), int(ninl[0])                   # t shape [3] (Q10)


class EssentialMatrixMetricSolver(EssentialMatrixSolver):
    '''E-mat decomposition + RANSAC for the translation scale from depth (pose_solver.py:115-172)'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_scale_threshold = cfg.EMAT_RANSAC.SCALE_THRESHOLD
        self._scale = ops.ScaleFromDepthBatch(self.ransac_scale_threshold)

    def estimate_pose(self, kpts0, kpts1, data):
        if len(kpts0) < 5:
            return _nan_pose()
        d, out = self._run(kpts0, kpts1, data, need_depth=True)
        # the scale stage is launched unconditionally (it propagates a failed E-mat status itself): no host round trip in between
        sc = self._scale(d["pts0"], d["pts1"], out["mask"], d["n"], d["depth0"], d["depth1"], d["K0"], d["K1"], out["R"], out["t"], out["status"])
        est, sst, R, tm, ninl, mask = self._stage.fetch(out["status"], sc["status"], out["R"], sc["t_metric"], sc["n_inliers"],
                                                       out["mask"][0, :d["n_host"]])
        self.mask = mask.astype(np.uint8).reshape(-1, 1)
        if int(est[0]) != ops.ST_OK or int(sst[0]) != ops.ST_OK:             # :131-132, :145-149
            return _nan_pose()
        return R.reshape(3, 3), tm.reshape(3, 1), int(ninl[0])


class ProcrustesSolver(_Base):
    '''Estimate relative pose (metric) using 3D-3D correspondences (pose_solver.py:238-320), incl. the optional
    whole-cloud ICP refinement (PROCRUSTES.REFINE, :290-315; csrc/procrustes_icp.hip)'''

    def __init__(self, cfg):
        super().__init__(cfg)
        self.ransac_max_corr_distance = cfg.PROCRUSTES.MAX_CORR_DIST
        self.refine = cfg.PROCRUSTES.REFINE
        self._solver = ops.ProcrustesBatchSolver(self.ransac_max_corr_distance, 0.999, self.seed)
        self._icp = ops.ProcrustesIcpRefine(self.ransac_max_corr_distance, 1e-4, 1e-4, 30) if self.refine else None     # :302-304

    def estimate_pose(self, pts0, pts1, data):
        if len(pts0) < 3:                                                   # :252-253
            return _nan_pose()
        d = self._stage.put(pts0, pts1, data, need_depth1=True)
        out = self._solver(d["pts0"], d["pts1"], d["n"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pid"])
        if self._icp is not None:                                            # :290-319, launched back to back: no host round trip
            ref = self._icp(d["depth0"], d["depth1"], d["K0"], d["K1"], out["R"], out["t"], out["status"])
            out = dict(out, n_inliers=ref["n_inliers"])
        st, R, t, ninl = self._stage.fetch(out["status"], out["R"], out["t"], out["n_inliers"])
        if int(st[0]) != ops.ST_OK:
            return _nan_pose()
        return R.reshape(3, 3), t.reshape(3, 1), int(ninl[0])
