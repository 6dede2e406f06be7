"""Synthetic 540x720 image pairs (no Map-free data offline): a band-limited random
texture seen from two views related by a small homography, so that a detector/descriptor finds
repeatable structure and the matcher has real correspondences to recover.  Values are float32 in
[0,1] like the reference's read_image (grayscale / 255, matchers.py:101-104)."""
import numpy as np


def _texture(rng, H, W):
    """sum of bilinearly up-sampled noise octaves + sparse blobs -> [H,W] float32 in [0,1]"""
    from scipy import ndimage
    img = np.zeros((H, W), np.float64)
    for cell, amp in ((64, 0.35), (24, 0.3), (9, 0.25), (4, 0.15)):
        h, w = H // cell + 2, W // cell + 2
        n = rng.uniform(0, 1, size=(h, w))
        img += amp * ndimage.zoom(n, (H / h, W / w), order=1)[:H, :W]
    img = (img - img.min()) / (img.max() - img.min())
    return img.astype(np.float32)


BAND_SHIFTS = (16, 8, 24)       # pixel disparity of the three horizontal depth bands (multiples of 8)


def synthetic_pair(seed, H=720, W=540, f=590.0, hard=False):
    """Two views of a scene made of three fronto-parallel depth bands (horizontal strips), the
    second camera translated along +x by tx: a point at depth Z moves by f*tx/Z pixels, and the
    band depths are chosen so that the disparities are exactly 16 / 8 / 24 px.  Disparities that
    are multiples of the networks' 8-px cell make SuperPoint's response exactly translation-
    equivariant inside each band even with untrained (random) weights, so the matcher has hundreds
    of true correspondences and the pose has a known answer.

    hard=True adds what makes the solvers work for their answer: independently MOVING objects (rectangles of the first view
    re-drawn in the second one displaced by (+-16 .. 32, +-16 .. 32) pixels, again multiples of 8 so the untrained networks
    still match them confidently) covering 30-55 % of the image.  Their correspondences are real matches but violate the
    epipolar geometry / the depth of the static scene: 30-60 % OUTLIERS for the E-matrix and PnP RANSACs, which then need
    hundreds of hypotheses and exercise the adaptive termination, the polish and the cheirality vote -- plus an occluder
    strip of unrelated texture in the second view (no matches, a depth discontinuity).  `outlier_area` = moving-object area.
    hard=2 additionally corrupts the DEPTH maps the way a monocular depth estimate is wrong (the reference lifts matches through
    DPT / KITTI-finetuned depth, lib/datasets/mapfree.py:150-163): 36-px blocks covering 35-55 % of each map are scaled by a
    factor in [1.2, 1.8] or its inverse, independently per view.  The images do not change, so the matchers return the same
    correspondences -- but 35-55 % of the lifted 3-D points are now wrong: outliers for PnP, Procrustes and the scale RANSAC
    whatever the matcher does.  `depth_outlier_frac` = corrupted fraction of depth0.

    returns dict(img0, img1 [H,W] f32 in [0,1], depth0, depth1 [H,W] f32 metres (uint16-mm
    quantised like lib/datasets/utils.py:77-81), K [3,3] f64, R_gt, t_gt)."""
    rng = np.random.default_rng(seed)
    pad = 32
    base = _texture(rng, H, W + 2 * pad)
    img0 = base[:, pad:pad + W].copy()
    img1 = np.empty_like(img0)
    depth = np.empty((H, W), np.float64)
    Za = 5.0
    tx = BAND_SHIFTS[0] * Za / f
    edges = [0, H // 3, 2 * H // 3, H]
    for k, sft in enumerate(BAND_SHIFTS):
        y0, y1 = edges[k], edges[k + 1]
        img1[y0:y1] = base[y0:y1, pad - sft:pad - sft + W]      # x1 = x0 + shift
        depth[y0:y1] = f * tx / sft
    depth1 = depth.copy()
    outlier_area = 0.0
    if hard:
        # occluder: a vertical strip of unrelated texture in the second view, in front of everything
        ow = 8 * int(rng.integers(3, 7)); ox = 8 * int(rng.integers(4, (W - ow) // 8 - 4))
        img1[:, ox:ox + ow] = _texture(rng, H, ow)
        depth1[:, ox:ox + ow] = 1.5
        # moving objects: content of view 0 re-drawn displaced in view 1
        area, target = 0, rng.uniform(0.30, 0.55) * H * W
        for _ in range(8):
            if area >= target:
                break
            hh = 8 * int(rng.integers(12, 36)); ww = 8 * int(rng.integers(10, 30))
            y0 = 8 * int(rng.integers(5, (H - hh) // 8 - 5)); x0 = 8 * int(rng.integers(5, (W - ww) // 8 - 5))
            dy = 8 * int(rng.choice([-4, -3, -2, 2, 3, 4])); dx = 8 * int(rng.choice([-4, -3, -2, 2, 3, 4]))
            img1[y0 + dy:y0 + dy + hh, x0 + dx:x0 + dx + ww] = img0[y0:y0 + hh, x0:x0 + ww]
            zo = float(rng.uniform(2.0, 4.0))
            depth[y0:y0 + hh, x0:x0 + ww] = zo
            depth1[y0 + dy:y0 + dy + hh, x0 + dx:x0 + dx + ww] = zo
            area += hh * ww
        outlier_area = float(min(area, H * W) / (H * W))
    depth_outlier_frac = 0.0
    if int(hard) >= 2:
        r2 = np.random.default_rng(seed + 77777)       # own stream: hard=1 scenes are unchanged
        blk = 36
        gh, gw = H // blk, W // blk
        for k, dm in enumerate((depth, depth1)):
            sel = r2.random((gh, gw)) < r2.uniform(0.35, 0.55)
            fac = r2.uniform(1.2, 1.8, (gh, gw))
            fac = np.where(r2.random((gh, gw)) < 0.5, fac, 1.0 / fac)
            m = np.where(sel, fac, 1.0)
            dm[:gh * blk, :gw * blk] *= np.kron(m, np.ones((blk, blk)))
            if k == 0:
                depth_outlier_frac = float(sel.mean() * gh * gw * blk * blk / (H * W))
    depth1 = (np.round(depth1 * 1000).astype(np.uint16) / 1000.0).astype(np.float32)
    depth = (np.round(depth * 1000).astype(np.uint16) / 1000.0).astype(np.float32)
    # float64, the dtype every Map-free sample carries (correct_intrinsic_scale multiplies a float64 eye(3) into K:
    # lib/datasets/utils.py:117-130, lib/datasets/mapfree.py:50-52) -- the solvers evaluate inv(K) / the K-normalisation in it
    K = np.array([[f, 0, W / 2 - 0.5], [0, f, H / 2 - 0.5], [0, 0, 1]], dtype=np.float64)
    return dict(img0=img0, img1=img1, depth0=depth, depth1=depth1, K=K,
                R_gt=np.eye(3), t_gt=np.array([tx, 0.0, 0.0]), outlier_area=outlier_area, depth_outlier_frac=depth_outlier_frac)


def synthetic_batch(seeds, H=720, W=540, hard=False):
    """dict of stacked arrays: images interleaved [2B,1,H,W] (image 2p = reference view, 2p+1 =
    query view of pair p), depth0/depth1 [B,H,W], K0/K1 [B,3,3], R_gt, t_gt, pair_ids"""
    prs = [synthetic_pair(s, H, W, hard=hard) for s in seeds]
    B = len(prs)
    images = np.empty((2 * B, 1, H, W), np.float32)
    for i, p in enumerate(prs):
        images[2 * i, 0], images[2 * i + 1, 0] = p["img0"], p["img1"]
    return dict(images=images, depth0=np.stack([p["depth0"] for p in prs]), depth1=np.stack([p["depth1"] for p in prs]),
                K0=np.stack([p["K"] for p in prs]), K1=np.stack([p["K"] for p in prs]),
                R_gt=np.stack([p["R_gt"] for p in prs]), t_gt=np.stack([p["t_gt"] for p in prs]),
                pair_ids=np.asarray(seeds, np.int64))
