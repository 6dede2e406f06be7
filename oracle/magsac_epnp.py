"""CPU ORACLE EXTENSION (TEST / STUDY INFRASTRUCTURE ONLY -- never imported by the product package).

Restatements, from the published algorithms, of the two OpenCV steps the solver oracle (oracle/mfr_oracle*.c) and the HIP kernels
simplify:

  * MAGSAC++ model quality + sigma-consensus++ local optimisation for `cv.findEssentialMat(..., method=cv.USAC_MAGSAC)`
    (lib/models/matching/pose_solver.py:46-48).  Barath, Noskova, Ivashechkin, Matas, "MAGSAC++, a fast, reliable and accurate
    robust estimator", CVPR 2020: residuals are chi-distributed with n = 4 degrees of freedom around a model whose noise scale
    sigma is marginalised over (0, sigma_max]; k = 3.64 is the 0.99 quantile.  Point loss (eq. 6-8 of the paper, up to the common
    constant C(n) 2^((n-1)/2) / sigma_max):
        rho(r) = sigma_max^2 / 2 * gamma_lower((n+1)/2, r^2 / (2 sigma_max^2))
                 + r^2 / 4 * (Gamma_upper((n-1)/2, r^2 / (2 sigma_max^2)) - Gamma_upper((n-1)/2, k^2 / 2))      for r < k sigma_max
        rho(r) = rho(k sigma_max)                                                                                  otherwise
    and the IRLS weight of sigma-consensus++  w(r) = Gamma_upper((n-1)/2, r^2 / (2 sigma_max^2)) - Gamma_upper((n-1)/2, k^2 / 2).
    The model with the SMALLEST total loss wins; local optimisation = iteratively re-weighted least squares on the epipolar
    constraint with those weights, projected onto the essential manifold, kept while the loss decreases.
    opencv-python 4.8.0.74 (environment.yml:17) is not available offline: its exact constants (how `threshold` maps to
    sigma_max, its lookup-table quantisation of the gamma functions, LO iteration counts) are NOT reproduced -- sigma_max is set so
    that k * sigma_max equals the Sampson threshold the reference passes.  PARITY UNPINNED vs OpenCV; this module exists to MEASURE
    how far a MAGSAC++-style selection moves the consensus set and the pose away from the inlier-count scheme the oracle and
    the kernels use (tools/magsac_epnp_study.py -> profiles/r03_magsac_epnp_study.json).

  * EPnP for the non-minimal refit inside `cv.solvePnPRansac` (pose_solver.py:209-213).  Lepetit, Moreno-Noguer, Fua, "EPnP: An
    accurate O(n) solution to the PnP problem", IJCV 2009: four control points (centroid + principal directions), the 2n x 12
    system M x = 0, null-space combinations for N = 1, 2, 3 (N = 4 omitted), Gauss-Newton on the betas, absolute orientation.
"""
import numpy as np
from scipy import special

K_QUANTILE = 3.64          # 0.99 quantile of the chi distribution with 4 degrees of freedom (the paper's k)
DOF = 4


# ---------------------------------------------------------------------------------------------------------------- MAGSAC++
def _gl(a, x):
    """lower incomplete gamma function (un-normalised)"""
    return special.gammainc(a, x) * special.gamma(a)


def _gu(a, x):
    """upper incomplete gamma function (un-normalised)"""
    return special.gammaincc(a, x) * special.gamma(a)


def magsac_loss(r2, sigma_max):
    """per-point MAGSAC++ loss for squared residuals r2 (same units as sigma_max^2)"""
    r2 = np.asarray(r2, np.float64)
    s2 = 2.0 * sigma_max * sigma_max
    cut = (K_QUANTILE * sigma_max) ** 2
    gk = _gu((DOF - 1) / 2.0, K_QUANTILE * K_QUANTILE / 2.0)

    def rho(q):
        return sigma_max * sigma_max / 2.0 * _gl((DOF + 1) / 2.0, q / s2) + q / 4.0 * (_gu((DOF - 1) / 2.0, q / s2) - gk)
    out = np.where(r2 < cut, rho(np.minimum(r2, cut)), rho(np.array(cut)))
    return out


def magsac_weight(r2, sigma_max):
    r2 = np.asarray(r2, np.float64)
    s2 = 2.0 * sigma_max * sigma_max
    gk = _gu((DOF - 1) / 2.0, K_QUANTILE * K_QUANTILE / 2.0)
    w = _gu((DOF - 1) / 2.0, r2 / s2) - gk
    return np.where(r2 < (K_QUANTILE * sigma_max) ** 2, np.maximum(w, 0.0), 0.0)


def sampson2(E, x0, x1):
    """squared Sampson distance of normalised correspondences (the oracle's inlier test, mfr_oracle_emat.c)"""
    p0 = np.c_[x0, np.ones(len(x0))]; p1 = np.c_[x1, np.ones(len(x1))]
    Ex0 = p0 @ E.T; Etx1 = p1 @ E
    num = np.sum(p1 * Ex0, 1) ** 2
    den = Ex0[:, 0] ** 2 + Ex0[:, 1] ** 2 + Etx1[:, 0] ** 2 + Etx1[:, 1] ** 2
    return num / np.maximum(den, 1e-300)


def project_essential(E):
    U, s, Vt = np.linalg.svd(E)
    m = (s[0] + s[1]) / 2.0
    return U @ np.diag([m, m, 0.0]) @ Vt


def weighted_eight_point(x0, x1, w):
    """weighted linear estimate of E from the epipolar constraint x1^T E x0 = 0, projected onto the essential manifold"""
    p0 = np.c_[x0, np.ones(len(x0))]; p1 = np.c_[x1, np.ones(len(x1))]
    A = (p1[:, :, None] * p0[:, None, :]).reshape(-1, 9) * np.sqrt(w)[:, None]
    _, _, Vt = np.linalg.svd(A, full_matrices=False)
    return project_essential(Vt[-1].reshape(3, 3))


def sigma_consensus_pp(E, x0, x1, sigma_max, iters=10):
    """local optimisation: IRLS with the MAGSAC++ weights, kept while the total loss decreases"""
    best, best_loss = E, float(magsac_loss(sampson2(E, x0, x1), sigma_max).sum())
    cur = E
    for _ in range(iters):
        w = magsac_weight(sampson2(cur, x0, x1), sigma_max)
        if (w > 0).sum() < 8:
            break
        cur = weighted_eight_point(x0, x1, w)
        loss = float(magsac_loss(sampson2(cur, x0, x1), sigma_max).sum())
        if loss < best_loss - 1e-12 * abs(best_loss):
            best, best_loss = cur, loss
        else:
            break
    return best, best_loss


def recover_pose(E, x0, x1, mask):
    """cv.recoverPose restated: the four (R, t) decompositions, cheirality vote over the masked points; returns the winner and
    the mask of points in front of both cameras (pose_solver.py:54-60 semantics)"""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    Wm = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    cands = [(U @ Wm @ Vt, U[:, 2]), (U @ Wm @ Vt, -U[:, 2]), (U @ Wm.T @ Vt, U[:, 2]), (U @ Wm.T @ Vt, -U[:, 2])]
    idx = np.nonzero(mask)[0]
    best = None
    for R, t in cands:
        # triangulate (midpoint-free linear DLT per point, as OpenCV's triangulatePoints)
        P0 = np.c_[np.eye(3), np.zeros(3)]; P1 = np.c_[R, t]
        good = np.zeros(len(x0), bool)
        for i in idx:
            A = np.stack([x0[i, 0] * P0[2] - P0[0], x0[i, 1] * P0[2] - P0[1], x1[i, 0] * P1[2] - P1[0], x1[i, 1] * P1[2] - P1[1]])
            X = np.linalg.svd(A)[2][-1]
            if abs(X[3]) < 1e-300:
                continue
            X = X[:3] / X[3]
            good[i] = X[2] > 0 and (R @ X + t)[2] > 0
        if best is None or good.sum() > best[2].sum():
            best = (R, t, good)
    return best


# ---------------------------------------------------------------------------------------------------------------- EPnP
def epnp(X, u, K):
    """X [n,3] world points, u [n,2] pixels, K [3,3] -> (R, t) minimising the algebraic EPnP error, then Gauss-Newton on the betas"""
    X = np.asarray(X, np.float64); u = np.asarray(u, np.float64)
    n = len(X)
    c0 = X.mean(0)
    Xc = X - c0
    _, s, Vt = np.linalg.svd(Xc, full_matrices=False)
    scale = np.sqrt(np.maximum(s * s / n, 1e-300))
    C = np.vstack([c0, c0 + scale[0] * Vt[0], c0 + scale[1] * Vt[1], c0 + scale[2] * Vt[2]])          # control points
    # barycentric coordinates: X = alpha C, sum alpha = 1
    A = np.c_[C, np.ones(4)].T
    al = np.linalg.solve(A, np.c_[X, np.ones(n)].T).T                                                  # [n,4]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = al[:, j] * fx; M[0::2, 3 * j + 2] = al[:, j] * (cx - u[:, 0])
        M[1::2, 3 * j + 1] = al[:, j] * fy; M[1::2, 3 * j + 2] = al[:, j] * (cy - u[:, 1])
    _, _, Vm = np.linalg.svd(M.T @ M)
    V = Vm[::-1][:4]                                                                                   # 4 smallest right singular vectors, smallest first
    dw = np.array([np.sum((C[i] - C[j]) ** 2) for i in range(4) for j in range(i + 1, 4)])            # 6 control-point distances
    pairs = [(i, j) for i in range(4) for j in range(i + 1, 4)]

    def dist_terms(v):
        c = v.reshape(4, 3)
        return np.array([c[i] - c[j] for i, j in pairs])                                               # [6,3]

    def pose_from(x):
        Cc = x.reshape(4, 3)
        if (al @ Cc)[:, 2].mean() < 0:
            Cc = -Cc
        Xcam = al @ Cc
        a0, b0 = X.mean(0), Xcam.mean(0)
        H = (X - a0).T @ (Xcam - b0)
        U, _, Vt2 = np.linalg.svd(H)
        R = Vt2.T @ U.T
        if np.linalg.det(R) < 0:
            Vt2[2] *= -1; R = Vt2.T @ U.T
        t = b0 - R @ a0
        Y = (R @ X.T).T + t
        pr = np.c_[fx * Y[:, 0] / Y[:, 2] + cx, fy * Y[:, 1] / Y[:, 2] + cy]
        return R, t, float(np.sqrt(np.mean(np.sum((pr - u) ** 2, 1))))

    def refine(betas, nb):
        D = [dist_terms(V[k]) for k in range(nb)]
        b = np.array(betas, np.float64)
        for _ in range(10):
            d = sum(b[k] * D[k] for k in range(nb))                                                    # [6,3]
            f = np.sum(d * d, 1) - dw
            J = np.stack([2.0 * np.sum(d * D[k], 1) for k in range(nb)], 1)
            step = np.linalg.lstsq(J, -f, rcond=None)[0]
            b = b + step
            if np.abs(step).max() < 1e-12:
                break
        return b

    best = None
    # N = 1
    d1 = dist_terms(V[0])
    b1 = np.sum(np.sqrt(np.sum(d1 * d1, 1) * dw)) / np.sum(np.sum(d1 * d1, 1))
    for nb, init in ((1, [b1]), (2, [b1, 0.0]), (3, [b1, 0.0, 0.0])):
        b = refine(init, nb)
        x = sum(b[k] * V[k] for k in range(nb))
        R, t, err = pose_from(x)
        if best is None or err < best[2]:
            best = (R, t, err)
    return best
