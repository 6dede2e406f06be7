"""Synthetic two-view geometry with known answers: used by tests and bench.py.

There is no Map-free data offline, so solver inputs are generated: 3-D points in a frustum,
a random relative pose (rotation <= 30 deg, |t| in [0.2, 2] m), pinhole projection with
K ~ [[590,0,269.5],[0,590,359.5],[0,0,1]] (540x720 images, config/mapfree.yaml:7-8), pixel
noise, a fraction of uniform outliers, and a depth map that carries the true depth (plus noise)
at the integer pixel under every keypoint of image 0 / image 1, stored like the dataset does
(uint16 millimetres / 1000, lib/datasets/utils.py:77-81).
"""
import numpy as np

H_MAPFREE, W_MAPFREE = 720, 540


def rand_rot(rng, maxdeg=30.0):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = np.deg2rad(rng.uniform(0, maxdeg))
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


def make_K(rng, H=H_MAPFREE, W=W_MAPFREE, jitter=0.05):
    f = 590.0 * W / 540.0 * (1 + rng.uniform(-jitter, jitter))
    return np.array([[f, 0, W / 2 - 0.5], [0, f * (1 + rng.uniform(-0.01, 0.01)), H / 2 - 0.5], [0, 0, 1]],
                    dtype=np.float32)


def make_pair(seed, n, outlier_frac=0.3, noise_px=1.0, H=H_MAPFREE, W=W_MAPFREE, depth_noise=0.0,
              zero_depth_frac=0.0):
    """Returns dict with pts0,pts1 [n,2] f32, depth0,depth1 [H,W] f32, K0,K1 [3,3] f32,
    R_gt [3,3], t_gt [3], inlier_gt [n] bool."""
    rng = np.random.default_rng(seed)
    K0, K1 = make_K(rng, H, W), make_K(rng, H, W)
    R = rand_rot(rng, 30.0)
    t = rng.normal(size=3)
    t *= rng.uniform(0.2, 2.0) / np.linalg.norm(t)
    pts0 = np.zeros((n, 2)); pts1 = np.zeros((n, 2)); z0 = np.zeros(n); z1 = np.zeros(n)
    filled = 0
    while filled < n:
        m = (n - filled) * 3 + 16
        u = rng.uniform(1, W - 2, m); v = rng.uniform(1, H - 2, m); z = rng.uniform(1.0, 10.0, m)
        X0 = np.stack([(u - K0[0, 2]) / K0[0, 0] * z, (v - K0[1, 2]) / K0[1, 1] * z, z], 1)
        X1 = (R @ X0.T).T + t
        u1 = K1[0, 0] * X1[:, 0] / X1[:, 2] + K1[0, 2]
        v1 = K1[1, 1] * X1[:, 1] / X1[:, 2] + K1[1, 2]
        ok = (X1[:, 2] > 0.3) & (u1 > 1) & (u1 < W - 2) & (v1 > 1) & (v1 < H - 2)
        k = min(int(ok.sum()), n - filled)
        sel = np.nonzero(ok)[0][:k]
        pts0[filled:filled + k] = np.stack([u[sel], v[sel]], 1)
        pts1[filled:filled + k] = np.stack([u1[sel], v1[sel]], 1)
        z0[filled:filled + k] = z[sel]; z1[filled:filled + k] = X1[sel, 2]
        filled += k
    pts1 = pts1 + rng.normal(size=(n, 2)) * noise_px
    inl = np.ones(n, dtype=bool)
    no = int(round(outlier_frac * n))
    if no > 0:
        oi = rng.permutation(n)[:no]
        pts1[oi] = np.stack([rng.uniform(1, W - 2, no), rng.uniform(1, H - 2, no)], 1)
        inl[oi] = False
    pts1[:, 0] = np.clip(pts1[:, 0], 0, W - 1.001); pts1[:, 1] = np.clip(pts1[:, 1], 0, H - 1.001)
    # depth maps: plane-ish background + true depth under each keypoint, quantised like the PNGs
    depth0 = np.full((H, W), 5.0); depth1 = np.full((H, W), 5.0)
    depth0 += rng.uniform(-0.5, 0.5, size=(H, W)); depth1 += rng.uniform(-0.5, 0.5, size=(H, W))
    p0i = pts0.astype(np.float32).astype(np.int32); p1i = pts1.astype(np.float32).astype(np.int32)
    depth0[p0i[:, 1], p0i[:, 0]] = z0 * (1 + depth_noise * rng.normal(size=n))
    depth1[p1i[inl, 1], p1i[inl, 0]] = (z1 * (1 + depth_noise * rng.normal(size=n)))[inl]
    if zero_depth_frac > 0:
        depth0[rng.uniform(size=(H, W)) < zero_depth_frac] = 0.0
        depth1[rng.uniform(size=(H, W)) < zero_depth_frac] = 0.0
    depth0 = (np.round(np.clip(depth0, 0, 65.0) * 1000).astype(np.uint16) / 1000.0).astype(np.float32)
    depth1 = (np.round(np.clip(depth1, 0, 65.0) * 1000).astype(np.uint16) / 1000.0).astype(np.float32)
    return dict(pts0=pts0.astype(np.float32), pts1=pts1.astype(np.float32), depth0=depth0, depth1=depth1,
                K0=K0, K1=K1, R_gt=R, t_gt=t, inlier_gt=inl)


def make_batch(seeds, n_list, maxN=None, **kw):
    """Stack pairs into the fixed-stride device layout of the C-ABI (numpy, host)."""
    pairs = [make_pair(s, n, **kw) for s, n in zip(seeds, n_list)]
    B = len(pairs)
    maxN = maxN or max(max(n_list), 1)
    H, W = pairs[0]["depth0"].shape
    out = dict(
        pts0=np.zeros((B, maxN, 2), np.float32), pts1=np.zeros((B, maxN, 2), np.float32),
        n_corr=np.array(n_list, np.int32),
        depth0=np.stack([p["depth0"] for p in pairs]), depth1=np.stack([p["depth1"] for p in pairs]),
        K0=np.stack([p["K0"] for p in pairs]), K1=np.stack([p["K1"] for p in pairs]),
        R_gt=np.stack([p["R_gt"] for p in pairs]), t_gt=np.stack([p["t_gt"] for p in pairs]),
        pair_ids=np.array(seeds, np.int64), pairs=pairs)
    for b, p in enumerate(pairs):
        out["pts0"][b, :n_list[b]] = p["pts0"]
        out["pts1"][b, :n_list[b]] = p["pts1"]
    return out


def rot_err_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.degrees(np.arccos(np.clip(c, -1, 1))))


def render_room_depth(H, W, f, R, t, noise=0.0, seed=0):
    """depth map (f32, metres, uint16-mm quantised like lib/datasets/utils.py:77-81) of an analytic scene -- floor, back wall,
    side wall and a slanted panel (four non-parallel planes n.X = c in the frame of camera 0) -- seen from the camera whose
    pose is X_cam = R X_0 + t.  Used to give the whole-cloud ICP refinement (pose_solver.py:290-315) a known answer."""
    planes = [(np.array([0.0, 1.0, 0.0]), 1.2), (np.array([0.0, 0.0, 1.0]), 4.0), (np.array([1.0, 0.0, 0.15]), 2.0),
              (np.array([-0.5, -0.3, 1.0]) / np.linalg.norm([-0.5, -0.3, 1.0]), 2.6)]
    K = np.array([[f, 0, W / 2 - 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]], dtype=np.float32)
    uu, vv = np.meshgrid(np.arange(W), np.arange(H))
    d = np.stack([(uu - K[0, 2]) / f, (vv - K[1, 2]) / f, np.ones_like(uu, dtype=np.float64)], -1)     # rays, z = 1, camera frame
    o = -R.T @ np.asarray(t, dtype=np.float64)                                                          # camera centre in frame 0
    dw = d @ R                                                                                          # rows: R^T d
    depth = np.full((H, W), np.inf)
    for n, c in planes:
        den = dw @ n
        lam = (c - o @ n) / np.where(np.abs(den) < 1e-12, np.nan, den)
        lam = np.where(lam > 0.05, lam, np.inf)
        depth = np.minimum(depth, lam)
    depth = np.where(np.isfinite(depth), depth, 0.0)
    if noise > 0:
        depth = depth * (1 + np.random.default_rng(seed).normal(0, noise, depth.shape)) * (depth > 0)
    depth = np.clip(depth, 0, 65.0)
    return (np.round(depth * 1000).astype(np.uint16) / 1000.0).astype(np.float32), K
