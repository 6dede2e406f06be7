"""How far does the ORACLE's own LoFTR + E-mat + scale pose move when its fine-level coordinates move by fp32 noise?

VERDICT r5 item 1(b): census pair seed 5007 / hard = 2 sat at 1.43e-4 m from the oracle (bar 1e-4 m) with the SAME 399 E-mat inliers
and match coordinates that differ in low bits.  This tool answers whether the bar is attainable at all for such a pair, from the CPU
oracle alone (no GPU, no product code in the measured path):

  (a) perturb mkpts1_f (the sub-pixel expectation, the only non-integer coordinates LoFTR emits: matchers.py:50-55) by uniform noise of
      amplitude d in {1e-6, 1e-5, 1e-4} px, K draws each, re-run the oracle's solver (pose_solver.py:125-172 restated in
      oracle/mfr_oracle*.c) and report the spread of (rotation, translation) against the unperturbed oracle pose;
  (a') flip k of the 2 M coordinates of mkpts1_f by ONE float32 ulp (k = 16 ... 512; the round-5 HIP path differed from the oracle in 186 of
      2 x 2294 coordinates on pair 5007, all in the last bits) -- the smallest change ANY other fp32 evaluation order produces;
  (b) (--thread-check) re-run the oracle's NETWORK with another intra-op thread count (torch.set_num_threads 1 vs N: the library picks other blockings /
      summation orders -- the fp32 noise of the reference against ITSELF) and report the coordinate and pose deltas.

Usage: python tools/loftr_sensitivity.py [--seeds 5007 5003] [--hard 2] [--draws 8] [--threads 8] [--thread-check] [--out gpurun_out/loftr_sensitivity.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd import images as IM  # noqa: E402
from oracle import pipeline_ref as PR, loftr_ref as LR, oracle_lib as O  # noqa: E402


def solve(pts, sb, i, pid, pix_thr=2.0, scale_thr=0.1, conf=0.9999):
    """the solver half of oracle/pipeline_ref.loftr_emat_pair on a given match list"""
    e = O.emat_solve(pts[:, :2], pts[:, 2:], sb["K0"][i], sb["K1"][i], pix_thr, conf, 1000, seed=0, pair_id=int(pid))
    if e["status"] != 0:
        return None
    sc = O.scale_lift(pts[:, :2], pts[:, 2:], e["mask"], sb["depth0"][i], sb["depth1"][i], sb["K0"][i], sb["K1"][i], e["R"], e["t"])
    if len(sc) == 0:
        return None
    cnt, bs, _ = O.scale_ransac(sc, scale_thr)
    return dict(R=e["R"], t=bs * e["t"], tdir=e["t"], scale=float(bs), n_emat=int(e["n_inl"]), n_scale=int(cnt), mask=e["mask"].astype(bool))


def pose_delta(a, b):
    c = np.clip((np.trace(a["R"].T @ b["R"]) - 1) / 2, -1, 1)
    cd = np.clip(float(np.dot(a["tdir"].reshape(3), b["tdir"].reshape(3))), -1, 1)
    return dict(rot_rad=float(np.arccos(c)), trans_m=float(np.linalg.norm(a["t"].reshape(3) - b["t"].reshape(3))),
                tdir_rad=float(np.arccos(cd)), scale_rel=float(abs(a["scale"] - b["scale"]) / max(abs(a["scale"]), 1e-12)),
                same_emat_mask=bool(np.array_equal(a["mask"], b["mask"])), n_scale=(a["n_scale"], b["n_scale"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[5007])
    ap.add_argument("--hard", type=int, default=2)
    ap.add_argument("--draws", type=int, default=8)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--thread-check", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "loftr_sensitivity.json"))
    a = ap.parse_args()
    model = PR._nets("loftr")
    res = dict(hard=a.hard, draws=a.draws, bar=dict(rot_rad=1e-4, trans_m=1e-4), pairs=[])
    for s in a.seeds:
        sb = IM.synthetic_batch([s], hard=a.hard)
        im = (PR._t(sb["images"][0, 0]), PR._t(sb["images"][1, 0]))
        t0 = time.perf_counter()
        torch.set_num_threads(a.threads)
        pts = LR.loftr_match_pair(model, *im)
        t_net = time.perf_counter() - t0
        base = solve(pts, sb, 0, s)
        rec = dict(seed=s, matches=int(len(pts)), net_seconds=round(t_net, 1))
        if base is None:
            rec["status"] = "oracle failed"
            res["pairs"].append(rec)
            continue
        rec.update(emat_inliers=base["n_emat"], inlier_fraction=round(base["n_emat"] / len(pts), 4), scale_inliers=base["n_scale"],
                   t_norm_m=float(np.linalg.norm(base["t"])))
        rng = np.random.default_rng(s)
        rec["perturbed"] = {}
        for amp in (1e-6, 1e-5, 1e-4):
            ds = []
            for _ in range(a.draws):
                q = pts.copy()
                q[:, 2:] = (q[:, 2:].astype(np.float64) + rng.uniform(-amp, amp, q[:, 2:].shape)).astype(np.float32)
                r = solve(q, sb, 0, s)
                if r is not None:
                    ds.append(pose_delta(base, r))
            rec["perturbed"][f"{amp:g}px"] = dict(
                solved=len(ds), max_rot_rad=max(d["rot_rad"] for d in ds), max_trans_m=max(d["trans_m"] for d in ds),
                median_trans_m=float(np.median([d["trans_m"] for d in ds])), max_tdir_rad=max(d["tdir_rad"] for d in ds),
                max_scale_rel=max(d["scale_rel"] for d in ds), same_emat_mask=sum(d["same_emat_mask"] for d in ds),
                beyond_bar=sum(d["rot_rad"] > 1e-4 or d["trans_m"] > 1e-4 for d in ds))
        rec["ulp_flips"] = {}
        for k in (16, 64, 186, 512):
            ds = []
            for _ in range(a.draws):
                q = pts.copy()
                flat = q[:, 2:].reshape(-1)
                idx = rng.choice(flat.size, size=min(k, flat.size), replace=False)
                up = rng.integers(0, 2, len(idx)).astype(bool)
                flat[idx] = np.where(up, np.nextafter(flat[idx], np.float32(np.inf)), np.nextafter(flat[idx], np.float32(-np.inf)))
                q[:, 2:] = flat.reshape(-1, 2)
                r = solve(q, sb, 0, s)
                if r is not None:
                    ds.append(pose_delta(base, r))
            same = [d for d in ds if d["same_emat_mask"]]
            rec["ulp_flips"][str(k)] = dict(
                solved=len(ds), beyond_bar=sum(d["rot_rad"] > 1e-4 or d["trans_m"] > 1e-4 for d in ds),
                same_emat_mask=len(same), max_rot_rad=max(d["rot_rad"] for d in ds), max_trans_m=max(d["trans_m"] for d in ds),
                median_trans_m=float(np.median([d["trans_m"] for d in ds])),
                max_trans_m_same_mask=max([d["trans_m"] for d in same], default=None), max_rot_rad_same_mask=max([d["rot_rad"] for d in same], default=None),
                max_tdir_rad_same_mask=max([d["tdir_rad"] for d in same], default=None), max_scale_rel_same_mask=max([d["scale_rel"] for d in same], default=None))
        if not a.thread_check:
            res["pairs"].append(rec)
            print(json.dumps(rec), flush=True)
            continue
        # (b) the reference network against itself at another thread count
        torch.set_num_threads(1 if a.threads > 1 else 2)
        t0 = time.perf_counter()
        pts1t = LR.loftr_match_pair(model, *im)
        rec["net_seconds_other_threads"] = round(time.perf_counter() - t0, 1)
        torch.set_num_threads(a.threads)
        ka = {(int(r[0]), int(r[1])): r for r in pts}
        kb = {(int(r[0]), int(r[1])): r for r in pts1t}
        common = sorted(set(ka) & set(kb))
        d = np.array([np.abs(ka[k] - kb[k]).max() for k in common]) if common else np.zeros(1)
        r = solve(pts1t, sb, 0, s)
        rec["oracle_vs_itself_other_thread_count"] = dict(
            threads=(a.threads, 1 if a.threads > 1 else 2), matches=(len(pts), len(pts1t)), common=len(common),
            coords_differ=int((d > 0).sum()), max_coord_px=float(d.max()), p99_coord_px=float(np.quantile(d, 0.99)),
            pose=pose_delta(base, r) if r is not None else None)
        res["pairs"].append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
