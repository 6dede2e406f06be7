"""How far would MAGSAC++ scoring + sigma-consensus++ (instead of inlier counting + one LM polish) and an EPnP refit (instead of the LM
refit) move the consensus sets and poses of the restated OpenCV solvers?    CPU only, oracle only.

For every pair of the SURVEY.md 8d known-answer sets (N in {256, 1024, 4096} x outliers {0.2, 0.5}, 1 px noise):
  E-mat   A = the oracle as shipped (== HIP kernels bit for bit): O.emat_solve.
          B = the SAME hypothesis stream (same Philox samples, same five-point solver) scored with the MAGSAC++ loss, the best model
              refined by sigma-consensus++ (IRLS), consensus set re-read at the reference's Sampson threshold, recoverPose restated.
          -> delta inlier count, Jaccard of the masks, delta rotation / translation direction, both errors against the truth.
  PnP     A = the oracle's pose (P3P RANSAC, LM refit, LM refinement).
          B = EPnP on A's RANSAC inlier set, then the same LM refinement (pose_solver.py:216-220).
          -> delta pose, reprojection inlier sets under both poses.
python tools/magsac_epnp_study.py [--pairs 8] [--out profiles/r03_magsac_epnp_study.json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd import synth  # noqa: E402
from oracle import oracle_lib as O, magsac_epnp as MG  # noqa: E402


def emat_magsac(p, seed, pix_thr=2.0, conf=0.9999, max_iters=1000):
    x0, x1 = O.normalize_points(p["pts0"], p["K0"]), O.normalize_points(p["pts1"], p["K1"])
    thr = O.emat_threshold(pix_thr, p["K0"], p["K1"])
    sigma_max = thr / MG.K_QUANTILE
    n = len(x0)
    best_loss, best_E, best_cnt = np.inf, None, 0
    niters, it = max_iters, 0
    while it < niters:
        s = O.sample_distinct(0, seed, it, n, 5)
        for E in O.fivept(x0[s], x1[s]):
            r2 = MG.sampson2(E, x0, x1)
            loss = float(MG.magsac_loss(r2, sigma_max).sum())
            if loss < best_loss:
                best_loss, best_E = loss, E
                cnt = int((r2 <= thr * thr).sum())
                if cnt > best_cnt:
                    best_cnt = cnt
                    niters = min(niters, O.update_num_iters(conf, 1.0 - cnt / n, 5, max_iters))
        it += 1
    if best_E is None:
        return None
    E, _ = MG.sigma_consensus_pp(best_E, x0, x1, sigma_max)
    mask = MG.sampson2(E, x0, x1) <= thr * thr
    R, t, good = MG.recover_pose(E, x0, x1, mask)
    return dict(R=R, t=t / np.linalg.norm(t), mask=good, ransac_mask=mask, iters=it)


def tdir_deg(a, b):
    a = a / np.linalg.norm(a); b = b / np.linalg.norm(b)
    return float(np.degrees(np.arccos(np.clip(a @ b, -1, 1))))


def reproj_inliers(R, t, xyz, obs, K, thr=3.0):
    Y = (R @ xyz.T).T + t.reshape(3)
    pr = np.c_[K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]]
    return (np.sum((pr - obs) ** 2, 1) <= thr * thr) & (Y[:, 2] > 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="profiles/r03_magsac_epnp_study.json")
    ap.add_argument("--pairs", type=int, default=8)
    a = ap.parse_args()
    res = []
    for n in (256, 1024, 4096):
        for outl in (0.2, 0.5):
            em, pn = [], []
            for k in range(a.pairs):
                seed = 7000 + 100 * n // 256 + int(outl * 10) * 17 + k
                p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
                tg = p["t_gt"] / np.linalg.norm(p["t_gt"])
                # ---- essential matrix
                A = O.emat_solve(p["pts0"], p["pts1"], p["K0"], p["K1"], 2.0, 0.9999, 1000, 0, seed)
                B = emat_magsac(p, seed)
                if A["status"] == 0 and B is not None:
                    ma, mb = A["mask"].astype(bool), B["mask"]
                    em.append(dict(d_count=int(mb.sum()) - int(ma.sum()), jaccard=float((ma & mb).sum() / max((ma | mb).sum(), 1)),
                                   d_rot_deg=synth.rot_err_deg(A["R"], B["R"]), d_tdir_deg=tdir_deg(A["t"].reshape(3), B["t"]),
                                   rotA=synth.rot_err_deg(A["R"], p["R_gt"]), rotB=synth.rot_err_deg(B["R"], p["R_gt"]),
                                   tA=tdir_deg(A["t"].reshape(3), tg), tB=tdir_deg(B["t"], tg),
                                   precA=float((ma & p["inlier_gt"]).sum() / max(ma.sum(), 1)), precB=float((mb & p["inlier_gt"]).sum() / max(mb.sum(), 1)),
                                   recA=float((ma & p["inlier_gt"]).sum() / p["inlier_gt"].sum()), recB=float((mb & p["inlier_gt"]).sum() / p["inlier_gt"].sum())))
                # ---- PnP
                xyz, obs, src = O.pnp_lift(p["pts0"], p["pts1"], p["depth0"], p["K0"])
                rr = O.pnp_ransac(xyz, obs, p["K1"], 1000, 3.0, 0.9999, 0, seed)
                if rr["status"] == 0 and rr["n_inl"] >= 6:
                    idx = np.nonzero(rr["mask"])[0].astype(np.int32)
                    K1 = p["K1"].astype(np.float64)
                    Re, te, _ = MG.epnp(xyz[idx], obs[idx], K1)
                    rc, Rb, tb = O.pnp_lm(xyz, obs, idx, p["K1"], Re, te)                     # pose_solver.py:216-220 on the EPnP start
                    Ra, ta = rr["R"], rr["t"].reshape(3)
                    ia, ib = reproj_inliers(Ra, ta, xyz, obs, K1), reproj_inliers(Rb, tb, xyz, obs, K1)
                    pn.append(dict(d_rot_deg=synth.rot_err_deg(Ra, Rb), d_trans_m=float(np.linalg.norm(ta - tb)),
                                   d_rot_epnp_only_deg=synth.rot_err_deg(Ra, Re), d_trans_epnp_only_m=float(np.linalg.norm(ta - te)),
                                   d_count=int(ib.sum()) - int(ia.sum()), jaccard=float((ia & ib).sum() / max((ia | ib).sum(), 1)),
                                   rotA=synth.rot_err_deg(Ra, p["R_gt"]), rotB=synth.rot_err_deg(Rb, p["R_gt"]),
                                   tA=float(np.linalg.norm(ta - p["t_gt"])), tB=float(np.linalg.norm(tb - p["t_gt"]))))
            q = lambda v, f=np.median: round(float(f(v)), 6) if len(v) else None
            rec = {"N": n, "outlier_frac": outl, "pairs": a.pairs}
            if em:
                rec["emat_magsacpp_vs_count"] = dict(
                    pairs=len(em), median_d_count=q([e["d_count"] for e in em]), max_abs_d_count=q([abs(e["d_count"]) for e in em], np.max),
                    median_jaccard=q([e["jaccard"] for e in em]), min_jaccard=q([e["jaccard"] for e in em], np.min),
                    median_d_rot_deg=q([e["d_rot_deg"] for e in em]), max_d_rot_deg=q([e["d_rot_deg"] for e in em], np.max),
                    median_d_tdir_deg=q([e["d_tdir_deg"] for e in em]), max_d_tdir_deg=q([e["d_tdir_deg"] for e in em], np.max),
                    rot_err_vs_truth=dict(count=q([e["rotA"] for e in em]), magsacpp=q([e["rotB"] for e in em])),
                    tdir_err_vs_truth=dict(count=q([e["tA"] for e in em]), magsacpp=q([e["tB"] for e in em])),
                    inlier_precision=dict(count=q([e["precA"] for e in em]), magsacpp=q([e["precB"] for e in em])),
                    inlier_recall=dict(count=q([e["recA"] for e in em]), magsacpp=q([e["recB"] for e in em])))
            if pn:
                rec["pnp_epnp_vs_lm_refit"] = dict(
                    pairs=len(pn), median_d_rot_deg=q([e["d_rot_deg"] for e in pn]), max_d_rot_deg=q([e["d_rot_deg"] for e in pn], np.max),
                    median_d_trans_m=q([e["d_trans_m"] for e in pn]), max_d_trans_m=q([e["d_trans_m"] for e in pn], np.max),
                    epnp_before_refinement=dict(median_d_rot_deg=q([e["d_rot_epnp_only_deg"] for e in pn]), median_d_trans_m=q([e["d_trans_epnp_only_m"] for e in pn])),
                    max_abs_d_count=q([abs(e["d_count"]) for e in pn], np.max), min_jaccard=q([e["jaccard"] for e in pn], np.min),
                    rot_err_vs_truth=dict(lm=q([e["rotA"] for e in pn]), epnp=q([e["rotB"] for e in pn])),
                    trans_err_vs_truth=dict(lm=q([e["tA"] for e in pn]), epnp=q([e["tB"] for e in pn])))
            print(json.dumps(rec), flush=True)
            res.append(rec)
    out = {"what": __doc__.split("\n\n")[0], "assumptions": "MAGSAC++ per the CVPR 2020 paper with k * sigma_max = the reference's Sampson threshold, n = 4; "
           "OpenCV 4.8's own constants / lookup tables / LO schedule are not available offline (parity unpinned vs OpenCV)", "sets": res}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
