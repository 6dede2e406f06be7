"""Matcher -> solver wire format of the reference: per scene one
`correspondences_{matcher}.npz` with key `correspondences`, float64 [Npairs, maxN, 4]
(x0,y0,x1,y1), NaN-padded, row order = all seq1 frames of poses.txt in file order.

stack_pts: etc/feature_matching_baselines/utils.py:59-69; writer: compute.py:84-85;
reader + NaN stripping: lib/models/matching/feature_matching.py:24-50; query-frame parsing:
utils.py:42-56.  Also the conversion between this format and the fixed-stride device layout
([B,maxN,2] x2 float32 + counts) the C-ABI consumes.
"""
import numpy as np


def stack_pts(pts_list):
    assert len(pts_list) > 0, 'list must not be empty'
    n = len(pts_list)
    max_npts = max(p.shape[0] for p in pts_list)
    d = pts_list[0].shape[1]
    out = np.full((n, max_npts, d), np.nan)
    for i, p in enumerate(pts_list):
        out[i, :p.shape[0]] = p
    return out


def save_correspondences(path, pts_list):
    np.savez_compressed(path, correspondences=stack_pts(pts_list))


def load_correspondences(path):
    return np.load(path, allow_pickle=False)['correspondences'].astype(np.float32)


def strip_nan(row):
    """one [maxN,4] row -> (pts1, pts2); empty -> (array([]), array([])) like the reference (Q12)"""
    corr = row[~np.isnan(row)].reshape(-1, 4)
    if len(corr) > 0:
        return corr[:, :2], corr[:, 2:]
    e = np.array([])
    return e, e


def device_batch_to_pts_list(pts0, pts1, n_corr):
    """device-layout numpy arrays -> list of [N,4] arrays (single NaN row when N == 0,
    matchers.py:59,120)"""
    out = []
    for b in range(len(n_corr)):
        n = int(n_corr[b])
        out.append(np.concatenate([pts0[b, :n], pts1[b, :n]], 1).astype(np.float64) if n > 0 else np.full((1, 4), np.nan))
    return out


def pts_rows_to_device_batch(rows, maxN=None):
    """list of NaN-padded [*,4] rows -> pts0, pts1 [B,maxN,2] float32, n_corr [B] int32"""
    stripped = [strip_nan(np.asarray(r, dtype=np.float32)) for r in rows]
    ns = [len(a) for a, _ in stripped]
    maxN = maxN or max(max(ns), 1)
    B = len(rows)
    p0 = np.zeros((B, maxN, 2), np.float32); p1 = np.zeros((B, maxN, 2), np.float32)
    for b, (a, c) in enumerate(stripped):
        if len(a):
            p0[b, :len(a)] = a; p1[b, :len(a)] = c
    return p0, p1, np.asarray(ns, np.int32)


def parse_mapfree_query_frames(pose_path):
    """all seq1 frames of poses.txt in file order (comments and the seq0 keyframe skipped)"""
    out = []
    with open(pose_path, 'r') as f:
        for l in f.readlines():
            if '#' in l or 'seq0' in l:
                continue
            out.append(l.strip().split(' ')[0])
    return out
