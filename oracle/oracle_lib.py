"""ctypes binding of the CPU oracle (oracle/mfr_oracle*.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmfr_oracle.so")
_SO_OVERRIDE = os.environ.get("MFR_ORACLE_SO", "")          # test infrastructure only: the sanitizer build (tests/test_oracle_sanitizers.py)

ST_OK, ST_TOO_FEW, ST_BAD_DEPTH, ST_NO_MODEL, ST_DEGENERATE = range(5)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("mfr_oracle.c", "mfr_oracle_emat.c", "mfr_oracle_procrustes.c", "mfr_oracle_icp.c", "mfr_oracle_desc.c", "mfr_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if _SO_OVERRIDE:
            _lib = C.CDLL(_SO_OVERRIDE)
        else:
            build()
            _lib = C.CDLL(_SO)
        _lib.mfr_ref_det_log.restype = C.c_double
        _lib.mfr_ref_det_log.argtypes = [C.c_double]
        _lib.mfr_ref_depth_min.restype = C.c_float
        _lib.mfr_ref_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


K_F32, K_F64 = 0, 1


def _K(K):
    """intrinsics in the dtype the caller holds them (float64 = the Map-free loader's flow, anything else float32) -> (array, tag)"""
    K = np.asarray(K)
    if K.dtype == np.float64:
        return np.ascontiguousarray(K).reshape(9), K_F64
    return np.ascontiguousarray(K, dtype=np.float32).reshape(9), K_F32


def _K2(K0, K1):
    """a pair of intrinsics in ONE dtype (float64 as soon as either is float64, as numpy promotion would)"""
    K0, K1 = np.asarray(K0), np.asarray(K1)
    if K0.dtype == np.float64 or K1.dtype == np.float64:
        K0, K1 = K0.astype(np.float64), K1.astype(np.float64)
    a, tag = _K(K0)
    b, _ = _K(K1)
    return a, b, tag


def philox(ctr, k0, k1):
    c = np.asarray(ctr, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().mfr_ref_philox4x32_10(_p(c), C.c_uint32(k0), C.c_uint32(k1), _p(out))
    return out


def sample_distinct(seed, pair_id, it, n, k):
    out = np.zeros(k, dtype=np.int32)
    lib().mfr_ref_sample_distinct(C.c_uint64(seed), C.c_uint64(pair_id), C.c_uint32(it), C.c_int(n), C.c_int(k), _p(out))
    return out


def det_log(x):
    return lib().mfr_ref_det_log(float(x))


def update_num_iters(p, ep, model_points, max_iters):
    return lib().mfr_ref_update_num_iters(float(p), float(ep), int(model_points), int(max_iters))


def poly_real_roots(c):
    c = _f64(c)
    out = np.zeros(16, dtype=np.float64)
    n = lib().mfr_ref_poly_real_roots(_p(c), C.c_int(len(c) - 1), _p(out))
    return out[:n].copy()


def backproject(uv, depth, K):
    uv = np.ascontiguousarray(uv, dtype=np.int32)
    depth = _f32(depth)
    K, kdt = _K(K)
    xyz = np.zeros((len(uv), 3), dtype=np.float64)
    rc = lib().mfr_ref_backproject(_p(uv), _p(depth), C.c_int(len(uv)), _p(K), C.c_int(kdt), _p(xyz))
    if rc:
        raise ValueError("unsupported K")
    return xyz


def pnp_lift(pts0, pts1, depth0, K0):
    pts0, pts1, depth0 = _f32(pts0), _f32(pts1), _f32(depth0)
    n = len(pts0)
    H, W = depth0.shape
    xyz = np.zeros((max(n, 1), 3)); obs = np.zeros((max(n, 1), 2)); src = np.zeros(max(n, 1), dtype=np.int32)
    K0, kdt = _K(K0)
    m = lib().mfr_ref_pnp_lift(_p(pts0), _p(pts1), C.c_int(n), _p(depth0), C.c_int(H), C.c_int(W),
                               _p(K0), C.c_int(kdt), _p(xyz), _p(obs), _p(src))
    if m < 0:
        raise ValueError("unsupported K")
    return xyz[:m].copy(), obs[:m].copy(), src[:m].copy()


def p3p(X, f):
    X, f = _f64(X), _f64(f)
    Rs = np.zeros((4, 3, 3)); ts = np.zeros((4, 3))
    n = lib().mfr_ref_p3p(_p(X), _p(f), _p(Rs), _p(ts))
    return Rs[:n].copy(), ts[:n].copy()


def pnp_ransac(xyz, obs, K1, max_iters=1000, thr=3.0, conf=0.9999, seed=0, pair_id=0, want_counts=False):
    xyz, obs = _f64(xyz), _f64(obs)
    n = len(xyz)
    R = np.zeros((3, 3)); t = np.zeros(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_inl = C.c_int(0); best_it = C.c_int(0); iters_run = C.c_int(0)
    counts = np.zeros(max_iters, dtype=np.int32) if want_counts else None
    K1, kdt = _K(K1)
    st = lib().mfr_ref_pnp_ransac(_p(xyz), _p(obs), C.c_int(n), _p(K1), C.c_int(kdt),
                                  C.c_int(max_iters), C.c_double(thr), C.c_double(conf),
                                  C.c_uint64(seed), C.c_uint64(pair_id), _p(R), _p(t), _p(mask),
                                  C.byref(n_inl), C.byref(best_it), C.byref(iters_run),
                                  _p(counts) if want_counts else None)
    out = dict(status=st, R=R, t=t, mask=mask[:n].copy(), n_inl=n_inl.value, best_iter=best_it.value,
               iters_run=iters_run.value)
    if want_counts:
        out["counts"] = counts
    return out


def pnp_solve(pts0, pts1, depth0, K0, K1, max_iters=1000, thr=3.0, conf=0.9999, seed=0, pair_id=0):
    pts0, pts1, depth0 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2), _f32(depth0)
    H, W = depth0.shape
    R = np.zeros((3, 3)); t = np.zeros(3); n_inl = C.c_int(0)
    K0, K1, kdt = _K2(K0, K1)
    st = lib().mfr_ref_pnp_solve(_p(pts0), _p(pts1), C.c_int(len(pts0)), _p(depth0), C.c_int(H), C.c_int(W),
                                 _p(K0), _p(K1), C.c_int(kdt), C.c_int(max_iters),
                                 C.c_double(thr), C.c_double(conf), C.c_uint64(seed), C.c_uint64(pair_id),
                                 _p(R), _p(t), C.byref(n_inl))
    return st, R, t.reshape(3, 1), n_inl.value


def pnp_lm(xyz, obs, idx, K1, R, t, max_iter=20):
    xyz, obs = _f64(xyz), _f64(obs)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    K1 = np.asarray(K1).reshape(9)
    Kd = np.array([K1[0], K1[4], K1[2], K1[5]], dtype=np.float64)
    R = _f64(R).copy(); t = _f64(t).reshape(3).copy()
    rc = lib().mfr_ref_pnp_lm(_p(xyz), _p(obs), _p(idx), C.c_int(len(idx)), _p(Kd), C.c_int(max_iter), _p(R), _p(t))
    return rc, R, t


def scale_lift(pts0, pts1, mask, depth0, depth1, K0, K1, R, t):
    pts0, pts1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    depth0, depth1 = _f32(depth0), _f32(depth1)
    H, W = depth0.shape
    n = len(pts0)
    mask_p = None
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        mask_p = _p(mask)
    scale = np.zeros(max(n, 1))
    K0, K1, kdt = _K2(K0, K1)
    m = lib().mfr_ref_scale_lift(_p(pts0), _p(pts1), mask_p, C.c_int(n), _p(depth0), _p(depth1),
                                 C.c_int(H), C.c_int(W), _p(K0), _p(K1), C.c_int(kdt),
                                 _p(_f64(R).reshape(9)), _p(_f64(t).reshape(3)), _p(scale))
    if m < 0:
        raise ValueError("unsupported K")
    return scale[:m].copy()


def scale_ransac(scale, thr):
    scale = _f64(scale)
    bs = C.c_double(0); bi = C.c_int(0)
    n = lib().mfr_ref_scale_ransac(_p(scale), C.c_int(len(scale)), C.c_double(thr), C.byref(bs), C.byref(bi))
    return n, bs.value, bi.value


def fivept(x0, x1):
    x0, x1 = _f64(x0).reshape(5, 2), _f64(x1).reshape(5, 2)
    Es = np.zeros((10, 3, 3))
    n = lib().mfr_ref_fivept(_p(x0), _p(x1), _p(Es))
    return Es[:n].copy()


def emat_decompose(E):
    E = _f64(E)
    Ra = np.zeros((3, 3)); Rb = np.zeros((3, 3)); t = np.zeros(3)
    rc = lib().mfr_ref_emat_decompose(_p(E), _p(Ra), _p(Rb), _p(t))
    return rc, Ra, Rb, t


def emat_threshold(pix_thr, K0, K1):
    lib().mfr_ref_emat_threshold.restype = C.c_double
    K0, K1, kdt = _K2(K0, K1)
    return lib().mfr_ref_emat_threshold(C.c_double(pix_thr), _p(K0), _p(K1), C.c_int(kdt))


def normalize_points(pts, K):
    pts = _f32(pts).reshape(-1, 2)
    out = np.zeros((len(pts), 2))
    K, kdt = _K(K)
    lib().mfr_ref_normalize_points(_p(pts), C.c_int(len(pts)), _p(K), C.c_int(kdt), _p(out))
    return out


EMAT_MAGSAC, EMAT_COUNT = 0, 1
MAGSAC_LUT_M = 2048
_LUT = {}


def magsac_lut(M=MAGSAC_LUT_M):
    """[M + 1, 2] table of (normalised MAGSAC++ loss, IRLS weight) over r^2 / cut in [0, 1] (mfr_ref_magsac_lut)"""
    if M not in _LUT:
        t = np.zeros((M + 1, 2))
        lib().mfr_ref_magsac_lut(_p(t), C.c_int(M))
        _LUT[M] = t
    return _LUT[M]


def orthonormalize(R):
    R = _f64(R).copy()
    lib().mfr_ref_orthonormalize(_p(R))
    return R


def emat_solve(pts0, pts1, K0, K1, pix_thr=2.0, conf=0.9999, max_iters=1000, seed=0, pair_id=0, want_counts=False,
               score=EMAT_MAGSAC, max_thr_ratio=1.0, lut_m=MAGSAC_LUT_M):
    pts0, pts1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    n = len(pts0)
    R = np.zeros((3, 3)); t = np.zeros(3)
    mask = np.zeros(max(n, 1), np.uint8); rmask = np.zeros(max(n, 1), np.uint8)
    n_inl = C.c_int(0); bi = C.c_int(0); ir = C.c_int(0); lo = C.c_int(0)
    counts = np.zeros(max_iters, np.int32) if want_counts else None
    losses = np.zeros(max_iters, np.float64) if want_counts else None
    K0, K1, kdt = _K2(K0, K1)
    lut = magsac_lut(lut_m)
    st = lib().mfr_ref_emat_solve(_p(pts0), _p(pts1), C.c_int(n), _p(K0), _p(K1), C.c_int(kdt),
                                  C.c_double(pix_thr), C.c_double(conf), C.c_int(max_iters), C.c_uint64(seed),
                                  C.c_uint64(pair_id), C.c_int(score), _p(lut), C.c_int(lut_m), C.c_double(max_thr_ratio),
                                  _p(R), _p(t), _p(mask), C.byref(n_inl), C.byref(bi), C.byref(ir),
                                  _p(counts) if want_counts else None, _p(losses) if want_counts else None, _p(rmask),
                                  C.byref(lo))
    out = dict(status=st, R=R, t=t, mask=mask[:n].copy(), ransac_mask=rmask[:n].copy(), n_inl=n_inl.value,
               best_iter=bi.value, iters_run=ir.value, lo_runs=lo.value)
    if want_counts:
        out["counts"] = counts
        out["losses"] = losses
    return out


def procrustes_lift(pts0, pts1, depth0, depth1, K0, K1):
    pts0, pts1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    depth0, depth1 = _f32(depth0), _f32(depth1)
    H, W = depth0.shape
    n = len(pts0)
    P = np.zeros((max(n, 1), 3)); Q = np.zeros((max(n, 1), 3))
    K0, K1, kdt = _K2(K0, K1)
    m = lib().mfr_ref_procrustes_lift(_p(pts0), _p(pts1), C.c_int(n), _p(depth0), _p(depth1), C.c_int(H), C.c_int(W),
                                      _p(K0), _p(K1), C.c_int(kdt), _p(P), _p(Q))
    return P[:m].copy(), Q[:m].copy()


def procrustes_ransac(P, Q, max_dist=0.05, conf=0.999, max_iters=4096, seed=0, pair_id=0, want_counts=False):
    P, Q = _f64(P), _f64(Q)
    n = len(P)
    R = np.zeros((3, 3)); t = np.zeros(3); n_inl = C.c_int(0); bi = C.c_int(0); ir = C.c_int(0)
    counts = np.zeros(max_iters, np.int32) if want_counts else None
    st = lib().mfr_ref_procrustes_ransac(_p(P), _p(Q), C.c_int(n), C.c_double(max_dist), C.c_double(conf), C.c_int(max_iters),
                                         C.c_uint64(seed), C.c_uint64(pair_id), _p(R), _p(t), C.byref(n_inl), C.byref(bi),
                                         C.byref(ir), _p(counts) if want_counts else None)
    out = dict(status=st, R=R, t=t, n_inl=n_inl.value, best_iter=bi.value, iters_run=ir.value)
    if want_counts:
        out["counts"] = counts
    return out


def procrustes_solve(pts0, pts1, depth0, depth1, K0, K1, max_dist=0.05, conf=0.999, max_iters=4096, seed=0, pair_id=0):
    pts0, pts1 = _f32(pts0).reshape(-1, 2), _f32(pts1).reshape(-1, 2)
    depth0, depth1 = _f32(depth0), _f32(depth1)
    H, W = depth0.shape
    R = np.zeros((3, 3)); t = np.zeros(3); n_inl = C.c_int(0)
    K0, K1, kdt = _K2(K0, K1)
    st = lib().mfr_ref_procrustes_solve(_p(pts0), _p(pts1), C.c_int(len(pts0)), _p(depth0), _p(depth1), C.c_int(H), C.c_int(W),
                                        _p(K0), _p(K1), C.c_int(kdt), C.c_double(max_dist), C.c_double(conf),
                                        C.c_int(max_iters), C.c_uint64(seed), C.c_uint64(pair_id), _p(R), _p(t), C.byref(n_inl))
    return st, R, t.reshape(3, 1), n_inl.value


def procrustes_icp(depth0, depth1, K0, K1, R, t, max_dist=0.05, rel_fitness=1e-4, rel_rmse=1e-4, max_iter=30):
    """pose_solver.py:290-319 (PROCRUSTES.REFINE): -> dict(R, t, n_inliers, fitness, rmse, iters)"""
    depth0, depth1 = _f32(depth0), _f32(depth1)
    H, W = depth0.shape
    R = _f64(R).reshape(9).copy(); t = _f64(t).reshape(3).copy()
    n_inl = C.c_int(0); fit = C.c_double(0); rmse = C.c_double(0)
    K0, K1, kdt = _K2(K0, K1)
    it = lib().mfr_ref_procrustes_icp(_p(depth0), _p(depth1), C.c_int(H), C.c_int(W), _p(K0), _p(K1), C.c_int(kdt),
                                      C.c_double(max_dist), C.c_double(rel_fitness), C.c_double(rel_rmse), C.c_int(max_iter), _p(R), _p(t),
                                      C.byref(n_inl), C.byref(fit), C.byref(rmse))
    return dict(R=R.reshape(3, 3), t=t, n_inliers=n_inl.value, fitness=fit.value, rmse=rmse.value, iters=it)


def rootsift(desc):
    """feature_matching.py:68-74 -> (rootSIFT descriptors f32 [n,128], squared norms f32 [n])"""
    desc = _f32(desc).reshape(-1, 128)
    out = np.zeros_like(desc); n2 = np.zeros(len(desc), np.float32)
    lib().mfr_ref_rootsift(_p(desc), C.c_int(len(desc)), _p(out), _p(n2))
    return out, n2


def desc_2nn(des0, des1, nrm0, nrm1):
    des0, des1 = _f32(des0).reshape(-1, 128), _f32(des1).reshape(-1, 128)
    n0, n1 = len(des0), len(des1)
    idx = np.zeros(max(n0, 1), np.int32); d2 = np.zeros((max(n0, 1), 2), np.float32)
    lib().mfr_ref_desc_2nn(_p(des0), _p(des1), _p(_f32(nrm0)), _p(_f32(nrm1)), C.c_int(n0), C.c_int(n1), _p(idx), _p(d2))
    return idx[:n0], d2[:n0]


def desc_ratio(nn_idx, nn_d2, n1, ratio, kp0, kp1):
    nn_idx = np.ascontiguousarray(nn_idx, np.int32); nn_d2 = _f32(nn_d2)
    kp0, kp1 = _f32(kp0).reshape(-1, 2), _f32(kp1).reshape(-1, 2)
    n0 = len(nn_idx)
    p0 = np.zeros((max(n0, 1), 2), np.float32); p1 = np.zeros((max(n0, 1), 2), np.float32)
    m = lib().mfr_ref_desc_ratio(_p(nn_idx), _p(nn_d2), C.c_int(n0), C.c_int(n1), C.c_double(ratio), _p(kp0), _p(kp1), _p(p0), _p(p1))
    return p0[:m].copy(), p1[:m].copy()


def sift_ratio_match(des0_raw, des1_raw, kp0, kp1, ratio=0.8):
    """whole a-3 leg after detectAndCompute: rootSIFT, exact 2-NN, ratio test -> (pts0, pts1)"""
    d0, n0 = rootsift(des0_raw); d1, n1 = rootsift(des1_raw)
    idx, d2 = desc_2nn(d0, d1, n0, n1)
    return desc_ratio(idx, d2, len(d1), ratio, kp0, kp1)
