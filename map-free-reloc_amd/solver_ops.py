"""Batched, device-resident pose-solver ops: thin torch<->C-ABI glue over csrc/pnp.hip,
csrc/scale.hip, csrc/emat.hip (include/mfr_hip.h).  All tensors live on the GPU; torch only
provides memory and the stream.

Reference leg being replaced: lib/models/matching/pose_solver.py (PnPSolver :175-235,
EssentialMatrixSolver :20-61, EssentialMatrixMetricSolver :115-172).
"""
import torch

from . import _lib

ST_OK, ST_TOO_FEW, ST_BAD_DEPTH, ST_NO_MODEL, ST_DEGENERATE = range(5)
NSEG = 16


def _chk(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.MfrLibraryError(f"{name} must be a CUDA(HIP) tensor: the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


K_F32, K_F64 = 0, 1        # include/mfr_hip.h MFR_K_F32 / MFR_K_F64


def _chk_K(K0, K1=None):
    """Intrinsics go to the library in the dtype the `data` dict holds them: float64 from the Map-free loader
    (lib/datasets/utils.py:117-130 multiplies a float64 eye(3) into K), float32 from resize=None datasets.  -> (K0, K1, tag);
    a mixed pair is promoted to float64, as numpy would promote the reference's arithmetic."""
    Ks = [K for K in (K0, K1) if K is not None]
    for K in Ks:
        if not (isinstance(K, torch.Tensor) and K.is_cuda):
            raise _lib.MfrLibraryError("K must be a CUDA(HIP) tensor: the HIP path has no CPU fallback")
        if K.dtype not in (torch.float32, torch.float64):
            raise TypeError(f"K: expected float32 or float64, got {K.dtype}")
    dt = torch.float64 if any(K.dtype == torch.float64 for K in Ks) else torch.float32
    out = [None if K is None else K.to(dt).contiguous() for K in (K0, K1)]
    return out[0], out[1], (K_F64 if dt == torch.float64 else K_F32)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def depth_min_partials(depth):
    """[B,H,W] f32 -> [B,16] partial minima (pose_solver.py:196 depth_0.min())."""
    lib = _lib.load(require_gpu=True)
    depth = _chk(depth, torch.float32, "depth")
    B, H, W = depth.shape
    out = torch.empty(B, NSEG, dtype=torch.float32, device=depth.device)
    _lib.check(lib.mfr_depth_min(_lib.ptr(depth), B, H, W, _lib.ptr(out), _lib.stream_ptr()), "mfr_depth_min")
    return out


def pnp_lift(pts0, pts1, n_corr, depth0, K0):
    """pose_solver.py:186-206: returns xyz [B,maxN,3] f64, obs [B,maxN,2] f64, src_idx, n_valid."""
    lib = _lib.load(require_gpu=True)
    pts0 = _chk(pts0, torch.float32, "pts0"); pts1 = _chk(pts1, torch.float32, "pts1")
    n_corr = _chk(n_corr, torch.int32, "n_corr"); depth0 = _chk(depth0, torch.float32, "depth0")
    K0, _, kdt = _chk_K(K0)
    B, maxN, _ = pts0.shape
    _, H, W = depth0.shape
    dev = pts0.device
    part = depth_min_partials(depth0)
    xyz = torch.zeros(B, maxN, 3, dtype=torch.float64, device=dev)
    obs = torch.zeros(B, maxN, 2, dtype=torch.float64, device=dev)
    src = torch.zeros(B, maxN, dtype=torch.int32, device=dev)
    nv = torch.zeros(B, dtype=torch.int32, device=dev)
    _lib.check(lib.mfr_pnp_lift(_lib.ptr(pts0), _lib.ptr(pts1), _lib.ptr(n_corr), B, maxN, _lib.ptr(depth0),
                                _lib.ptr(part), H, W, _lib.ptr(K0), kdt, _lib.ptr(xyz), _lib.ptr(obs), _lib.ptr(src),
                                _lib.ptr(nv), _lib.stream_ptr()), "mfr_pnp_lift")
    return xyz, obs, src, nv


def pnp_ransac(xyz, obs, n_valid, K1, pair_ids, max_iters=1000, thr=3.0, conf=0.9999, seed=0):
    """cv.solvePnPRansac(P3P) + refit + ITERATIVE refinement restatement (pose_solver.py:209-235)
    on already lifted points.  Returns dict(R, t, n_inliers, status, mask, best_iter, iters_run, counts)."""
    lib = _lib.load(require_gpu=True)
    xyz = _chk(xyz, torch.float64, "xyz"); obs = _chk(obs, torch.float64, "obs")
    n_valid = _chk(n_valid, torch.int32, "n_valid"); K1, _, kdt = _chk_K(K1)
    pair_ids = _chk(pair_ids, torch.int64, "pair_ids")
    B, maxN, _ = xyz.shape
    dev = xyz.device
    max_iters = max(int(max_iters), 1)
    counts = torch.empty(B, max_iters, dtype=torch.int32, device=dev)
    inl = torch.empty(B, maxN, dtype=torch.int32, device=dev)
    R = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
    t = torch.empty(B, 3, dtype=torch.float64, device=dev)
    ni = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    mask = torch.empty(B, maxN, dtype=torch.uint8, device=dev)
    bi = torch.empty(B, dtype=torch.int32, device=dev)
    ir = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(lib.mfr_pnp_ransac(_lib.ptr(xyz), _lib.ptr(obs), _lib.ptr(n_valid), B, maxN, _lib.ptr(K1), kdt, max_iters,
                                  float(thr), float(conf), int(seed), _lib.ptr(pair_ids), _lib.ptr(counts),
                                  _lib.ptr(inl), _lib.ptr(R), _lib.ptr(t), _lib.ptr(ni), _lib.ptr(st), _lib.ptr(mask),
                                  _lib.ptr(bi), _lib.ptr(ir), _lib.stream_ptr()), "mfr_pnp_ransac")
    return dict(R=R, t=t, n_inliers=ni, status=st, mask=mask, best_iter=bi, iters_run=ir, counts=counts)


class PnPBatchSolver:
    """PnPSolver.estimate_pose (pose_solver.py:184-235) for a batch of pairs, one C-ABI call."""

    def __init__(self, max_iters=1000, reproj_thr=3.0, confidence=0.9999, seed=0):
        self.max_iters = max(int(max_iters), 1)
        self.reproj_thr = float(reproj_thr)
        self.confidence = float(confidence)
        self.seed = int(seed)
        self._ws = None

    def __call__(self, pts0, pts1, n_corr, depth0, K0, K1, pair_ids, want_mask=False):
        lib = _lib.load(require_gpu=True)
        pts0 = _chk(pts0, torch.float32, "pts0"); pts1 = _chk(pts1, torch.float32, "pts1")
        n_corr = _chk(n_corr, torch.int32, "n_corr"); depth0 = _chk(depth0, torch.float32, "depth0")
        K0, K1, kdt = _chk_K(K0, K1)
        pair_ids = _chk(pair_ids, torch.int64, "pair_ids")
        B, maxN, _ = pts0.shape
        _, H, W = depth0.shape
        dev = pts0.device
        need = lib.mfr_pnp_workspace_bytes(B, maxN, self.max_iters)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = _ws(need, dev)
        R = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
        t = torch.empty(B, 3, dtype=torch.float64, device=dev)
        ni = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        mask = torch.empty(B, maxN, dtype=torch.uint8, device=dev) if want_mask else None
        _lib.check(lib.mfr_pnp_solve_batch(
            _lib.ptr(pts0), _lib.ptr(pts1), _lib.ptr(n_corr), B, maxN, _lib.ptr(depth0), H, W, _lib.ptr(K0),
            _lib.ptr(K1), kdt, self.max_iters, self.reproj_thr, self.confidence, self.seed, _lib.ptr(pair_ids),
            _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(R), _lib.ptr(t), _lib.ptr(ni), _lib.ptr(st),
            _lib.ptr(mask), _lib.stream_ptr()), "mfr_pnp_solve_batch")
        out = dict(R=R, t=t, n_inliers=ni, status=st)
        if want_mask:
            out["mask"] = mask
        return out


EMAT_SCORE = {"magsac": 0, "count": 1}      # MFR_EMAT_SCORE_* of include/mfr_hip.h
MAGSAC_LUT_M = 2048


def magsac_lut_host(M=MAGSAC_LUT_M):
    """[M + 1, 2] float64 (normalised MAGSAC++ loss, IRLS weight) over r^2 / cut in [0, 1]: mfr_magsac_lut (host code of the library)"""
    import numpy as np
    t = np.zeros((M + 1, 2), dtype=np.float64)
    _lib.check(_lib.load().mfr_magsac_lut(t.ctypes.data, M), "mfr_magsac_lut")
    return t


class EssentialBatchSolver:
    """EssentialMatrixSolver.estimate_pose (pose_solver.py:29-61) for a batch of pairs.  score 'magsac' (default) = what the
    reference asks OpenCV for (cv.USAC_MAGSAC, :46-48): MAGSAC++ model quality + sigma-consensus++; 'count' = inlier count + LM
    polish (rounds 1-3, kept for A/B)."""

    def __init__(self, pix_thr=2.0, confidence=0.9999, seed=0, max_iters=1000, score="magsac", max_thr_ratio=1.0):
        if score not in EMAT_SCORE:
            raise ValueError(f"EssentialBatchSolver: unknown score {score!r}; known: {sorted(EMAT_SCORE)}")
        if not max_thr_ratio >= 1.0:
            raise ValueError("EssentialBatchSolver: max_thr_ratio must be >= 1")
        self.pix_thr, self.confidence = float(pix_thr), float(confidence)
        self.seed, self.max_iters = int(seed), max(int(max_iters), 1)
        self.score, self.max_thr_ratio = EMAT_SCORE[score], float(max_thr_ratio)
        self._ws = None
        self._lut = None

    def _table(self, dev):
        if self._lut is None or self._lut.device != dev:
            self._lut = torch.from_numpy(magsac_lut_host()).to(dev)
        return self._lut

    def __call__(self, pts0, pts1, n_corr, K0, K1, pair_ids, diagnostics=False):
        lib = _lib.load(require_gpu=True)
        pts0 = _chk(pts0, torch.float32, "pts0"); pts1 = _chk(pts1, torch.float32, "pts1")
        n_corr = _chk(n_corr, torch.int32, "n_corr")
        K0, K1, kdt = _chk_K(K0, K1)
        pair_ids = _chk(pair_ids, torch.int64, "pair_ids")
        B, maxN, _ = pts0.shape
        dev = pts0.device
        need = lib.mfr_emat_workspace_bytes(B, maxN, self.max_iters)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = _ws(need, dev)
        lut = self._table(dev) if self.score == 0 else None
        R = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
        t = torch.empty(B, 3, dtype=torch.float64, device=dev)
        ni = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        mask = torch.empty(B, maxN, dtype=torch.uint8, device=dev)
        bi = ir = cnt = los = lo = None
        if diagnostics:
            bi = torch.empty(B, dtype=torch.int32, device=dev); ir = torch.empty(B, dtype=torch.int32, device=dev)
            lo = torch.empty(B, dtype=torch.int32, device=dev)
            cnt = torch.empty(B, self.max_iters, dtype=torch.int32, device=dev)
            los = torch.zeros(B, self.max_iters, dtype=torch.float64, device=dev)
        _lib.check(lib.mfr_emat_solve_batch(
            _lib.ptr(pts0), _lib.ptr(pts1), _lib.ptr(n_corr), B, maxN, _lib.ptr(K0), _lib.ptr(K1), kdt, self.pix_thr,
            self.confidence, self.max_iters, self.seed, _lib.ptr(pair_ids), self.score, _lib.ptr(lut), MAGSAC_LUT_M,
            self.max_thr_ratio, _lib.ptr(self._ws), self._ws.numel(),
            _lib.ptr(R), _lib.ptr(t), _lib.ptr(ni), _lib.ptr(st), _lib.ptr(mask), _lib.ptr(bi), _lib.ptr(ir),
            _lib.ptr(cnt), _lib.ptr(los), _lib.ptr(lo), _lib.stream_ptr()), "mfr_emat_solve_batch")
        out = dict(R=R, t=t, n_inliers=ni, status=st, mask=mask)
        if diagnostics:
            out.update(best_iter=bi, iters_run=ir, counts=cnt, losses=los, lo_runs=lo)
        return out


class ProcrustesBatchSolver:
    """ProcrustesSolver.estimate_pose (pose_solver.py:238-320, REFINE False) for a batch of pairs."""

    def __init__(self, max_corr_dist=0.05, confidence=0.999, seed=0, max_iters=4096):
        self.max_corr_dist, self.confidence = float(max_corr_dist), float(confidence)
        self.seed, self.max_iters = int(seed), max(int(max_iters), 1)
        self._ws = None

    def __call__(self, pts0, pts1, n_corr, depth0, depth1, K0, K1, pair_ids, diagnostics=False):
        lib = _lib.load(require_gpu=True)
        pts0 = _chk(pts0, torch.float32, "pts0"); pts1 = _chk(pts1, torch.float32, "pts1")
        n_corr = _chk(n_corr, torch.int32, "n_corr")
        depth0 = _chk(depth0, torch.float32, "depth0"); depth1 = _chk(depth1, torch.float32, "depth1")
        K0, K1, kdt = _chk_K(K0, K1)
        pair_ids = _chk(pair_ids, torch.int64, "pair_ids")
        B, maxN, _ = pts0.shape
        _, H, W = depth0.shape
        dev = pts0.device
        need = lib.mfr_procrustes_workspace_bytes(B, maxN, self.max_iters)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = _ws(need, dev)
        R = torch.empty(B, 3, 3, dtype=torch.float64, device=dev)
        t = torch.empty(B, 3, dtype=torch.float64, device=dev)
        ni = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        bi = ir = cnt = None
        if diagnostics:
            bi = torch.empty(B, dtype=torch.int32, device=dev); ir = torch.empty(B, dtype=torch.int32, device=dev)
            cnt = torch.empty(B, self.max_iters, dtype=torch.int32, device=dev)
        _lib.check(lib.mfr_procrustes_solve_batch(
            _lib.ptr(pts0), _lib.ptr(pts1), _lib.ptr(n_corr), B, maxN, _lib.ptr(depth0), _lib.ptr(depth1), H, W, _lib.ptr(K0),
            _lib.ptr(K1), kdt, self.max_corr_dist, self.confidence, self.max_iters, self.seed, _lib.ptr(pair_ids),
            _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(R), _lib.ptr(t), _lib.ptr(ni), _lib.ptr(st), _lib.ptr(bi),
            _lib.ptr(ir), _lib.ptr(cnt), _lib.stream_ptr()), "mfr_procrustes_solve_batch")
        out = dict(R=R, t=t, n_inliers=ni, status=st)
        if diagnostics:
            out.update(best_iter=bi, iters_run=ir, counts=cnt)
        return out


class ProcrustesIcpRefine:
    """PROCRUSTES.REFINE (pose_solver.py:290-319): whole-cloud point-to-point ICP from the RANSAC transform, for a batch of
    pairs (csrc/procrustes_icp.hip).  R [B,3,3], t [B,3] f64 are refined IN PLACE; returns n_inliers / fitness / rmse / iters."""

    def __init__(self, max_corr_dist=0.05, rel_fitness=1e-4, rel_rmse=1e-4, max_iter=30):
        self.max_corr_dist, self.rel_fitness, self.rel_rmse = float(max_corr_dist), float(rel_fitness), float(rel_rmse)
        self.max_iter = int(max_iter)
        self._ws = None

    def __call__(self, depth0, depth1, K0, K1, R, t, status=None):
        lib = _lib.load(require_gpu=True)
        depth0 = _chk(depth0, torch.float32, "depth0"); depth1 = _chk(depth1, torch.float32, "depth1")
        K0, K1, kdt = _chk_K(K0, K1)
        R = _chk(R, torch.float64, "R"); t = _chk(t, torch.float64, "t")
        if status is not None:
            status = _chk(status, torch.int32, "status")
        B, H, W = depth0.shape
        dev = depth0.device
        need = lib.mfr_procrustes_icp_workspace_bytes(B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = _ws(need, dev)
        ni = torch.empty(B, dtype=torch.int32, device=dev)
        fit = torch.empty(B, dtype=torch.float64, device=dev); rm = torch.empty(B, dtype=torch.float64, device=dev)
        it = torch.empty(B, dtype=torch.int32, device=dev)
        _lib.check(lib.mfr_procrustes_icp_refine(_lib.ptr(depth0), _lib.ptr(depth1), B, H, W, _lib.ptr(K0), _lib.ptr(K1), kdt, self.max_corr_dist,
                                                 self.rel_fitness, self.rel_rmse, self.max_iter, _lib.ptr(status), _lib.ptr(self._ws),
                                                 self._ws.numel(), _lib.ptr(R), _lib.ptr(t), _lib.ptr(ni), _lib.ptr(fit), _lib.ptr(rm),
                                                 _lib.ptr(it), _lib.stream_ptr()), "mfr_procrustes_icp_refine")
        return dict(R=R, t=t, n_inliers=ni, fitness=fit, rmse=rm, iters=it)


class ScaleFromDepthBatch:
    """EssentialMatrixMetricSolver's own part (pose_solver.py:137-172) for a batch of pairs."""

    def __init__(self, scale_thr=0.1):
        self.scale_thr = float(scale_thr)
        self._ws = None

    def __call__(self, pts0, pts1, emat_mask, n_corr, depth0, depth1, K0, K1, R, t, in_status=None):
        lib = _lib.load(require_gpu=True)
        pts0 = _chk(pts0, torch.float32, "pts0"); pts1 = _chk(pts1, torch.float32, "pts1")
        n_corr = _chk(n_corr, torch.int32, "n_corr")
        depth0 = _chk(depth0, torch.float32, "depth0"); depth1 = _chk(depth1, torch.float32, "depth1")
        K0, K1, kdt = _chk_K(K0, K1)
        R = _chk(R, torch.float64, "R"); t = _chk(t, torch.float64, "t")
        if emat_mask is not None:
            emat_mask = _chk(emat_mask, torch.uint8, "emat_mask")
        if in_status is not None:
            in_status = _chk(in_status, torch.int32, "in_status")
        B, maxN, _ = pts0.shape
        _, H, W = depth0.shape
        dev = pts0.device
        need = lib.mfr_scale_workspace_bytes(B, maxN)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = _ws(need, dev)
        tm = torch.empty(B, 3, dtype=torch.float64, device=dev)
        bs = torch.empty(B, dtype=torch.float64, device=dev)
        ni = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        _lib.check(lib.mfr_scale_from_depth_batch(
            _lib.ptr(pts0), _lib.ptr(pts1), _lib.ptr(emat_mask), _lib.ptr(n_corr), B, maxN, _lib.ptr(depth0),
            _lib.ptr(depth1), H, W, _lib.ptr(K0), _lib.ptr(K1), kdt, _lib.ptr(R), _lib.ptr(t), _lib.ptr(in_status),
            self.scale_thr, _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(tm), _lib.ptr(bs), _lib.ptr(ni),
            _lib.ptr(st), _lib.stream_ptr()), "mfr_scale_from_depth_batch")
        return dict(t_metric=tm, best_scale=bs, n_inliers=ni, status=st)


def test_f64_ops(a, b, c):
    lib = _lib.load(require_gpu=True)
    a = _chk(a, torch.float64, "a"); b = _chk(b, torch.float64, "b"); c = _chk(c, torch.float64, "c")
    out = torch.empty(a.numel(), 3, dtype=torch.float64, device=a.device)
    _lib.check(lib.mfr_test_f64_ops(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), a.numel(), _lib.ptr(out),
                                    _lib.stream_ptr()), "mfr_test_f64_ops")
    return out


def test_sample(seed, pair_ids, iters, n, k):
    lib = _lib.load(require_gpu=True)
    pair_ids = _chk(pair_ids, torch.int64, "pair_ids")
    B = pair_ids.numel()
    out = torch.empty(B, iters, k, dtype=torch.int32, device=pair_ids.device)
    _lib.check(lib.mfr_test_sample(int(seed), _lib.ptr(pair_ids), B, iters, n, k, _lib.ptr(out), _lib.stream_ptr()),
               "mfr_test_sample")
    return out
