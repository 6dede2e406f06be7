"""Solver quality on the SURVEY.md 8d known-answer sets (N in {256, 1024, 4096} x outliers {0.2, 0.5}, 1 px noise, 16 pairs each):
pose error against the ground truth and inlier-set precision / recall against the generated inlier mask, for the restated
PnP-RANSAC and E-mat-RANSAC (CPU oracle == HIP kernels bit for bit).  This is the evidence available offline for the
simplified OpenCV steps (inlier counting instead of MAGSAC++ sigma-marginalisation, LM instead of EPnP for the non-minimal
refit): what their consensus sets and poses look like against the truth.  python tools/known_answer_stats.py [--out profiles/r02_known_answer_stats.json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd import synth  # noqa: E402
from oracle import oracle_lib as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="profiles/r02_known_answer_stats.json")
    ap.add_argument("--pairs", type=int, default=16)
    a = ap.parse_args()
    res = []
    for n in (256, 1024, 4096):
        for outl in (0.2, 0.5):
            rec = {"N": n, "outlier_frac": outl, "pairs": a.pairs}
            pr, pt, er, et, prec, recl, ppr, prc, iters = [], [], [], [], [], [], [], [], []
            for k in range(a.pairs):
                seed = 7000 + 100 * n // 256 + int(outl * 10) * 17 + k
                p = synth.make_pair(seed, n, outlier_frac=outl, noise_px=1.0, depth_noise=0.002)
                st, R, t, ninl = O.pnp_solve(p["pts0"], p["pts1"], p["depth0"], p["K0"], p["K1"], 1000, 3.0, 0.9999, 0, seed)
                if st == 0:
                    pr.append(synth.rot_err_deg(R, p["R_gt"])); pt.append(float(np.linalg.norm(t.ravel() - p["t_gt"])))
                    xyz, obs, src = O.pnp_lift(p["pts0"], p["pts1"], p["depth0"], p["K0"])
                    rr = O.pnp_ransac(xyz, obs, p["K1"], 1000, 3.0, 0.9999, 0, seed)
                    m = np.zeros(n, bool); m[src[rr["mask"].astype(bool)]] = True
                    ppr.append(float((m & p["inlier_gt"]).sum() / max(m.sum(), 1))); prc.append(float((m & p["inlier_gt"]).sum() / p["inlier_gt"].sum()))
                    iters.append(rr["iters_run"])
                e = O.emat_solve(p["pts0"], p["pts1"], p["K0"], p["K1"], 2.0, 0.9999, 1000, 0, seed)
                if e["status"] == 0:
                    er.append(synth.rot_err_deg(e["R"], p["R_gt"]))
                    tg = p["t_gt"] / np.linalg.norm(p["t_gt"])
                    et.append(float(np.degrees(np.arccos(np.clip(float(e["t"].reshape(3) @ tg), -1, 1)))))
                    m = e["mask"].astype(bool)
                    prec.append(float((m & p["inlier_gt"]).sum() / max(m.sum(), 1))); recl.append(float((m & p["inlier_gt"]).sum() / p["inlier_gt"].sum()))
            med = lambda v: round(float(np.median(v)), 5) if v else None
            rec["pnp"] = dict(solved=len(pr), median_rot_deg=med(pr), median_trans_m=med(pt), inlier_precision=med(ppr), inlier_recall=med(prc),
                              median_iterations=med(iters))
            rec["emat"] = dict(solved=len(er), median_rot_deg=med(er), median_tdir_deg=med(et), inlier_precision=med(prec), inlier_recall=med(recl))
            print(json.dumps(rec), flush=True)
            res.append(rec)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
