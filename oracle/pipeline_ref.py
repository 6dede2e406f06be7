"""Whole-path CPU oracle: the reference's per-pair flow restated end to end, one pair at a time.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline / parity legs,
tools/parity_census.py).  The product package never imports this module.

  sg_pnp_pair       SuperGlue_matcher.match (etc/feature_matching_baselines/matchers.py:93-120)
                    -> PrecomputedMatching (lib/models/matching/feature_matching.py:28-50)
                    -> PnPSolver.estimate_pose (lib/models/matching/pose_solver.py:184-235)
                    [config/matching/mapfree/sg_pnp_dptkitti.yaml]
  loftr_emat_pair   LoFTR_matcher.match (matchers.py:24-59, incl. the 544 padding quirk Q3)
                    -> EssentialMatrixMetricSolver.estimate_pose (pose_solver.py:125-172)
                    [config/matching/mapfree/loftr_emat_dptkitti.yaml]

Networks: oracle/nets_ref.py, oracle/loftr_ref.py (PyTorch-CPU fp32); solvers: oracle/mfr_oracle*.c.
"""
import numpy as np
import torch

from . import nets_ref as NR, loftr_ref as LR, oracle_lib as O

_NETS = {}


def _nets(kind):
    """seeded synthetic weights of the product package (the only weights available offline)"""
    if kind not in _NETS:
        from mapfree_reloc_amd.nets import weights as WT
        if kind == "sg":
            sp = NR.SuperPointRef().eval(); sp.load_state_dict(WT.superpoint_state_dict())
            sg = NR.SuperGlueRef().eval(); sg.load_state_dict(WT.superglue_state_dict())
            _NETS[kind] = (sp, sg)
        else:
            m = LR.LoFTRRef().eval(); m.load_state_dict(WT.loftr_state_dict())
            _NETS[kind] = m
    return _NETS[kind]


def _t(img):
    return torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]


def _nan_result(pts, extra=None):
    out = dict(pts=pts, status=1, R=np.full((3, 3), np.nan), t=np.full(3, np.nan), n_inliers=0)
    out.update(extra or {})
    return out


@torch.no_grad()
def sg_pnp_pair(img0, img1, depth0, K0, K1, pair_id, pnp_iters=1000, pnp_thr=3.0, pnp_conf=0.9999, seed=0):
    """-> dict(pts [N,4] (single NaN row when empty), status, R [3,3], t [3], n_inliers)"""
    sp, sg = _nets("sg")
    pts = NR.superglue_match_pair(sp, sg, _t(img0), _t(img1))
    if np.isnan(pts).any():                                     # feature_matching.py:44-48 -> pose_solver.py:188
        return _nan_result(pts)
    st, R, t, ninl = O.pnp_solve(pts[:, :2], pts[:, 2:], depth0, K0, K1, pnp_iters, pnp_thr, pnp_conf, seed=seed, pair_id=int(pair_id))
    if st != 0:
        return _nan_result(pts, dict(status=int(st)))
    # the inlier INDEX set (over the correspondences handed to the solver): the same solve, stage by stage, to get at the mask
    xyz, obs, src = O.pnp_lift(pts[:, :2], pts[:, 2:], depth0, K0)
    rr = O.pnp_ransac(xyz, obs, K1, pnp_iters, pnp_thr, pnp_conf, seed=seed, pair_id=int(pair_id))
    mask = np.zeros(len(pts), bool)
    mask[src[rr["mask"].astype(bool)]] = True
    assert int(mask.sum()) == int(ninl) and np.array_equal(rr["R"], R)
    return dict(pts=pts, status=0, R=R, t=t.reshape(3), n_inliers=int(ninl), mask=mask)


@torch.no_grad()
def loftr_emat_pair(img0, img1, depth0, depth1, K0, K1, pair_id, pix_thr=2.0, scale_thr=0.1, conf=0.9999, seed=0):
    """-> dict(pts, status, R, t (metric), n_inliers (= scale inliers, the submission confidence), emat_inliers, mask)"""
    model = _nets("loftr")
    pts = LR.loftr_match_pair(model, _t(img0), _t(img1))
    if np.isnan(pts).any() or len(pts) < 5:                     # pose_solver.py:32-33
        return _nan_result(pts, dict(emat_inliers=0))
    e = O.emat_solve(pts[:, :2], pts[:, 2:], K0, K1, pix_thr, conf, 1000, seed=seed, pair_id=int(pair_id))
    if e["status"] != 0:
        return _nan_result(pts, dict(status=int(e["status"]), emat_inliers=0))
    sc = O.scale_lift(pts[:, :2], pts[:, 2:], e["mask"], depth0, depth1, K0, K1, e["R"], e["t"])
    if len(sc) == 0:                                            # :145-149
        return _nan_result(pts, dict(status=2, emat_inliers=int(e["n_inl"]), mask=e["mask"]))
    cnt, bs, _ = O.scale_ransac(sc, scale_thr)
    return dict(pts=pts, status=0, R=e["R"], t=bs * e["t"], n_inliers=int(cnt), emat_inliers=int(e["n_inl"]), mask=e["mask"])


@torch.no_grad()
def sg_procrustes_pair(img0, img1, depth0, depth1, K0, K1, pair_id, max_dist=0.05, seed=0):
    """SuperGlue matches -> ProcrustesSolver.estimate_pose, PROCRUSTES.REFINE False (pose_solver.py:238-320;
    config/matching/mapfree/sg_procrustes_dptkitti.yaml)"""
    sp, sg = _nets("sg")
    pts = NR.superglue_match_pair(sp, sg, _t(img0), _t(img1))
    if np.isnan(pts).any():
        return _nan_result(pts)
    st, R, t, ninl = O.procrustes_solve(pts[:, :2], pts[:, 2:], depth0, depth1, K0, K1, max_dist, 0.999, 4096, seed=seed, pair_id=int(pair_id))
    if st != 0:
        return _nan_result(pts, dict(status=int(st)))
    return dict(pts=pts, status=0, R=R, t=t.reshape(3), n_inliers=int(ninl))


def inlier_rows(pts, mask):
    """canonical form of an inlier index set: the sorted (x0, y0, x1, y1) rows of the inliers (independent of match order)"""
    p = np.asarray(pts, np.float32).reshape(-1, 4)[np.asarray(mask, bool)[:len(pts)]]
    return p[np.lexsort(p.T[::-1])] if len(p) else p


def compare_pair(ref, hip_pts, hip_R, hip_t, hip_ninl, hip_status, hip_mask=None):
    """per-pair census record: is the match set identical (bit-equal coordinates, same order), what fraction of
    the oracle's matches the HIP path also has, the pose delta (rad / m) and the inlier-count delta"""
    rp = ref["pts"]
    rn = 0 if np.isnan(rp).any() else len(rp)
    hp = np.asarray(hip_pts, dtype=np.float32).reshape(-1, 4)
    same = rn == len(hp) and (rn == 0 or np.array_equal(rp.astype(np.float32), hp))
    # set overlap on exact (x0,y0,x1,y1) rows
    rs = {tuple(r) for r in rp.astype(np.float32).tolist()} if rn else set()
    hs = {tuple(r) for r in hp.tolist()}
    both = len(rs & hs)
    # coarse identity (same keypoint pairing after rounding to 1/64 px: fp32 sub-pixel noise of LoFTR's fine stage)
    rq = {tuple(np.round(np.asarray(r) * 64).astype(np.int64).tolist()) for r in rs}
    hq = {tuple(np.round(np.asarray(r) * 64).astype(np.int64).tolist()) for r in hs}
    rec = dict(n_ref=rn, n_hip=len(hp), identical_matches=bool(same), identical_set=bool(rs == hs), common_exact=both, common_q64=len(rq & hq),
               status_ref=int(ref["status"]), status_hip=int(hip_status), inliers_ref=int(ref["n_inliers"]), inliers_hip=int(hip_ninl))
    if hip_mask is not None and "mask" in ref and ref["status"] == 0 and hip_status == 0:
        a, b2 = inlier_rows(rp, ref["mask"]), inlier_rows(hp, hip_mask)
        rec["inlier_set_identical"] = bool(a.shape == b2.shape and np.array_equal(a, b2))
        sa, sb = {tuple(r) for r in a.tolist()}, {tuple(r) for r in b2.tolist()}
        rec["inlier_set_jaccard"] = float(len(sa & sb) / max(len(sa | sb), 1))
        # the same at 1/64 px (LoFTR's fine level is a sub-pixel fp32 expectation: coordinates agree to ~1e-5 px, not bit for bit)
        qa = {tuple(np.round(np.asarray(r) * 64).astype(np.int64).tolist()) for r in sa}
        qb = {tuple(np.round(np.asarray(r) * 64).astype(np.int64).tolist()) for r in sb}
        rec["inlier_set_identical_q64"] = bool(qa == qb)
        rec["inlier_set_jaccard_q64"] = float(len(qa & qb) / max(len(qa | qb), 1))
        rec["inlier_fraction_ref"] = float(len(a) / max(rn, 1))
    if ref["status"] == 0 and hip_status == 0:
        Rr, Rh = np.asarray(ref["R"], dtype=np.float64), np.asarray(hip_R, dtype=np.float64).reshape(3, 3)
        c = np.clip((np.trace(Rr.T @ Rh) - 1) / 2, -1, 1)
        rec["rot_rad"] = float(np.arccos(c))
        rec["trans_m"] = float(np.linalg.norm(np.asarray(ref["t"]).reshape(3) - np.asarray(hip_t).reshape(3)))
        rec["pose_bit_equal"] = bool(np.array_equal(Rr, Rh) and np.array_equal(np.asarray(ref["t"]).reshape(3), np.asarray(hip_t).reshape(3)))
    return rec


def summarize(records):
    """census summary published in bench.py's config.parity and asserted by tests/test_gpu_parity_census.py"""
    n = len(records)
    ident = [r for r in records if r["identical_matches"]]
    posed = [r for r in records if "rot_rad" in r]
    out = dict(pairs=n, identical_match_sets=len(ident), identical_fraction=round(len(ident) / max(n, 1), 4),
               identical_as_sets=sum(r.get("identical_set", False) for r in records),
               status_agree=sum(r["status_ref"] == r["status_hip"] for r in records),
               mean_common_fraction=round(float(np.mean([r["common_q64"] / max(r["n_ref"], 1) for r in records])), 5) if n else None,
               inlier_count_equal=sum(r["inliers_ref"] == r["inliers_hip"] for r in records))
    if posed:
        out.update(max_rot_rad=float(max(r["rot_rad"] for r in posed)), max_trans_m=float(max(r["trans_m"] for r in posed)),
                   median_rot_rad=float(np.median([r["rot_rad"] for r in posed])), median_trans_m=float(np.median([r["trans_m"] for r in posed])),
                   pose_within_bar=sum(r["rot_rad"] <= 1e-4 and r["trans_m"] <= 1e-4 for r in posed), posed_pairs=len(posed),
                   pose_bit_equal=sum(r.get("pose_bit_equal", False) for r in posed))
    ins = [r for r in records if "inlier_set_identical" in r]
    if ins:
        out.update(inlier_index_sets_compared=len(ins), inlier_index_sets_identical=sum(r["inlier_set_identical"] for r in ins),
                   min_inlier_set_jaccard=round(float(min(r["inlier_set_jaccard"] for r in ins)), 5),
                   inlier_index_sets_identical_q64=sum(r["inlier_set_identical_q64"] for r in ins),
                   min_inlier_set_jaccard_q64=round(float(min(r["inlier_set_jaccard_q64"] for r in ins)), 5),
                   median_inlier_fraction=round(float(np.median([r["inlier_fraction_ref"] for r in ins])), 4),
                   min_inlier_fraction=round(float(min(r["inlier_fraction_ref"] for r in ins)), 4))
    pi = [r for r in ident if "rot_rad" in r]
    if pi:
        out.update(identical_max_rot_rad=float(max(r["rot_rad"] for r in pi)), identical_max_trans_m=float(max(r["trans_m"] for r in pi)))
    return out
