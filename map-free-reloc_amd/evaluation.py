"""Local evaluation = the metric definition of the headline numbers (BASELINE.json "median rot/trans
err on val"): restatement of the reference's benchmark code -- per-frame translation error, rotation
error (sin variant, quirk Q14), virtual-correspondence reprojection error (VCRE), per-scene medians
averaged over scenes, precision and AUC over the confidence-sorted precision/recall curve.

Follows benchmark/metrics.py:40-67, benchmark/reprojection.py:7-86, benchmark/utils.py:12-182,
benchmark/mapfree.py:17-117, benchmark/config.py:3-8.  transforms3d is not installed offline, so the
quaternion algebra it provides (w,x,y,z convention) is written out here.  CPU numpy post-processing:
outside the accelerated path, pinned by fixtures produced by the reference's own code
(oracle/gen_golden.py -> tests/golden/ref_metrics.npz).
"""
from collections import defaultdict

import numpy as np

T_THRESHOLD, R_THRESHOLD, VCRE_THRESHOLD = 0.25, 5, 90          # benchmark/config.py:3-8


# ---- quaternion algebra (transforms3d.quaternions, w-first) ----
def qmult(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def qinverse(q):
    q = np.asarray(q, dtype=np.float64)
    return np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)


def quat2mat(q):
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ, xX, xY, xZ, yY, yZ, zZ = w * X, w * Y, w * Z, x * X, x * Y, x * Z, y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def rotate_vector(v, q):
    return qmult(q, qmult(np.r_[0.0, v], np.array([q[0], -q[1], -q[2], -q[3]])))[1:]


def convert_world2cam_to_cam2world(q, t):
    qinv = qinverse(q)
    return qinv, -rotate_vector(t, qinv)


# ---- per-frame metrics ----
def trans_err(t_est, t_gt):
    return np.linalg.norm(np.asarray(t_est) - np.asarray(t_gt))


def rot_err(q_est, q_gt):
    """sin variant: arcsin(|vec(q_gt_n * q_est_n^-1)|) * 2 * 180 / pi"""
    q1 = np.asarray(q_gt, np.float64) / np.linalg.norm(q_gt)
    q2 = np.asarray(q_est, np.float64) / np.linalg.norm(q_est)
    sine = qmult(q1, qinverse(q2))
    return float(np.arcsin(np.linalg.norm(sine[1:])) * 114.59155902616465)


def project(pts, K, img_size=None):
    uv_h = (K @ pts[:, :3].T).T
    uv = uv_h[:, :2] / uv_h[:, -1:]
    if img_size is not None:
        uv[:, 0] = np.clip(uv[:, 0], 0, img_size[0]); uv[:, 1] = np.clip(uv[:, 1], 0, img_size[1])
    return uv


def _vcre_grid():
    step = 0.3
    xs = (np.arange(0, 7) - 3.0) * step
    ys = (np.arange(0, 4) - 1.5) * step
    zs = np.arange(0, 7).astype(float) * step + 1.8
    xx, yy, zz = np.meshgrid(xs, ys, zs)
    return np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1), np.ones(xx.size)], -1)


_GRID = _vcre_grid()


def reproj_err(q_est, t_est, q_gt, t_gt, K, W, H):
    uv_gt = project(_GRID, K, (W, H))
    A = np.eye(4); A[:3, :3] = quat2mat(q_est); A[:3, -1] = t_est
    G = np.eye(4); G[:3, :3] = quat2mat(q_gt); G[:3, -1] = t_gt
    res = (np.linalg.inv(A) @ G @ _GRID.T).T
    uv = project(res, K, (W, H))
    return float(np.linalg.norm(uv_gt - uv, ord=2, axis=1).mean())


def frame_metrics(q_est, t_est, confidence, q_gt, t_gt, K, W, H):
    return dict(trans_err=trans_err(t_est, t_gt), rot_err=rot_err(q_est, q_gt),
                reproj_err=reproj_err(q_est, t_est, q_gt, t_gt, K, W, H), confidence=confidence)


# ---- aggregation ----
def precision_recall(inliers, tp, failures):
    inliers = np.array(inliers)
    order = np.argsort(inliers)[::-1]
    inliers = inliers[order]
    tp = np.array(tp).reshape(-1)[order]
    thr_idx = np.r_[np.where(np.diff(inliers))[0], inliers.size - 1]
    N = inliers.shape[0]
    rec = np.arange(N, dtype=np.float32) + 1
    prec = np.cumsum(tp)[thr_idx] / rec[thr_idx]
    rec = rec[thr_idx] / (float(N) + float(failures))
    last = rec.searchsorted(rec[-1])
    sl = slice(last, None, -1)
    prec = np.r_[prec[sl], 1]; rec = np.r_[rec[sl], 0]
    return prec, rec, np.abs(np.sum(np.diff(rec) * np.array(prec)[:-1]))


def aggregate_results(all_results, all_failures):
    """all_results: {scene: {metric: [values]}} -> the benchmark's headline dictionary"""
    med, allm = defaultdict(list), defaultdict(list)
    for sr in all_results.values():
        for m, v in sr.items():
            med[m].append(np.median(v)); allm[m].extend(v)
    allm = {k: np.array(v) for k, v in allm.items()}
    avg_med = {m: np.mean(v) for m, v in med.items()}
    ok_pose = (allm['trans_err'] < T_THRESHOLD) * (allm['rot_err'] < R_THRESHOLD)
    ok_vcre = allm['reproj_err'] < VCRE_THRESHOLD
    total = len(next(iter(allm.values()))) + all_failures
    _, _, auc_pose = precision_recall(allm['confidence'], ok_pose, all_failures)
    _, _, auc_vcre = precision_recall(allm['confidence'], ok_vcre, all_failures)
    return {
        'Average Median Translation Error': avg_med['trans_err'],
        'Average Median Rotation Error': avg_med['rot_err'],
        'Average Median Reprojection Error': avg_med['reproj_err'],
        f'Precision @ Pose Error < ({T_THRESHOLD*100}cm, {R_THRESHOLD}deg)': np.sum(ok_pose) / total,
        f'AUC @ Pose Error < ({T_THRESHOLD*100}cm, {R_THRESHOLD}deg)': auc_pose,
        f'Precision @ VCRE < {VCRE_THRESHOLD}px': np.sum(ok_vcre) / total,
        f'AUC @ VCRE < {VCRE_THRESHOLD}px': auc_vcre,
        'Estimates for % of frames': len(allm['trans_err']) / total,
    }
