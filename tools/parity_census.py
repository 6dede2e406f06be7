"""End-to-end parity census: the whole HIP pipeline vs the whole CPU-oracle pipeline on N synthetic pairs per configuration --
per pair: match set identical?, INLIER INDEX SET identical (canonical order)?, pose delta, inlier-count delta -- on the easy
3-band scenes and on the HARD ones (moving objects + occluder: 30-60 % outliers, images.synthetic_pair(hard=True)).
Configurations: sg_pnp (configs[1]), loftr_emat (configs[2]), sg_procrustes (f-1), sift_emat (configs[0]: descriptor leg -> E-mat).
Usage: python tools/parity_census.py [--sg 64] [--loftr 16] [--procrustes 16] [--sift 32] [--hard 2] [--out gpurun_out/census.json]"""
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
from mapfree_reloc_amd import images as IM, synth  # noqa: E402
from mapfree_reloc_amd import solver_ops as ops  # noqa: E402
from mapfree_reloc_amd.pipeline import LoFTREmatPipeline, SuperGluePnPPipeline  # noqa: E402
from oracle import pipeline_ref as PR, oracle_lib as O  # noqa: E402


def _truth(rec, sb, i, R, t, status):
    if int(status) == 0:
        c = np.clip((np.trace(sb["R_gt"][i].T @ np.asarray(R, np.float64).reshape(3, 3)) - 1) / 2, -1, 1)
        rec["rot_err_vs_truth_deg"] = float(np.degrees(np.arccos(c)))
        rec["trans_err_vs_truth_m"] = float(np.linalg.norm(np.asarray(t).reshape(3) - sb["t_gt"][i]))


def census(kind, seeds, dev="cuda", chunk=8, threads=16, hard=False):
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    if kind == "sift_emat":
        return census_sift(seeds, dev)
    pipe = LoFTREmatPipeline(dev) if kind == "loftr_emat" else SuperGluePnPPipeline(dev)
    proc = ops.ProcrustesBatchSolver(0.05, 0.999, 0) if kind == "sg_procrustes" else None
    recs = []
    for lo in range(0, len(seeds), chunk):
        ss = seeds[lo:lo + chunk]
        sb = IM.synthetic_batch(ss, hard=hard)
        d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb.items() if isinstance(v, np.ndarray)}
        if kind == "sg_pnp":
            out = pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"], want_mask=True)
        elif kind == "sg_procrustes":
            m = pipe.match(d["images"])
            out = proc(m["pts0"], m["pts1"], m["n_corr"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pair_ids"])
            out.update(n_corr=m["n_corr"], pts0=m["pts0"], pts1=m["pts1"])
        else:
            out = pipe(d["images"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pair_ids"])
        torch.cuda.synchronize()
        o = {k: v.cpu().numpy() for k, v in out.items() if isinstance(v, torch.Tensor)}
        for i, s in enumerate(ss):
            a = (sb["images"][2 * i, 0], sb["images"][2 * i + 1, 0])
            if kind == "sg_pnp":
                ref = PR.sg_pnp_pair(*a, sb["depth0"][i], sb["K0"][i], sb["K1"][i], s)
                hmask = o["mask"][i]
            elif kind == "sg_procrustes":
                ref = PR.sg_procrustes_pair(*a, sb["depth0"][i], sb["depth1"][i], sb["K0"][i], sb["K1"][i], s)
                hmask = None
            else:
                ref = PR.loftr_emat_pair(*a, sb["depth0"][i], sb["depth1"][i], sb["K0"][i], sb["K1"][i], s)
                hmask = o["emat_mask"][i]
            n = int(o["n_corr"][i])
            r = PR.compare_pair(ref, np.concatenate([o["pts0"][i, :n], o["pts1"][i, :n]], 1), o["R"][i], o["t"][i], o["n_inliers"][i], o["status"][i],
                                hip_mask=None if hmask is None else hmask[:n])
            r["seed"] = int(s)
            _truth(r, sb, i, o["R"][i], o["t"][i], o["status"][i])
            recs.append(r)
    return recs


def _sift_like(rng, n):
    d = rng.gamma(0.6, 1.0, (n, 128))
    d = np.minimum(d / np.linalg.norm(d, axis=1, keepdims=True), 0.2)
    return np.clip(np.rint(512.0 * d / np.linalg.norm(d, axis=1, keepdims=True)), 0, 255).astype(np.float32)


def census_sift(seeds, dev):
    """configs[0] without cv2: keypoints = a synthetic two-view geometry (synth.make_pair: 1 px noise, 40 % wrong correspondences),
    descriptors = SIFT-like vectors, the second view's a noisy copy for the true matches -> rootSIFT + exact 2-NN + ratio test
    (HIP: csrc/descriptor_match.hip, oracle: mfr_oracle_desc.c) -> E-mat RANSAC (HIP: csrc/emat.hip, oracle: mfr_oracle_emat.c)"""
    from mapfree_reloc_amd import descriptor_ops as DO
    em = ops.EssentialBatchSolver(2.0, 0.9999, 0)
    recs = []
    for s in seeds:
        n = 2048
        p = synth.make_pair(9000 + s, n, outlier_frac=0.4, noise_px=1.0)
        rng = np.random.default_rng(s)
        d0 = _sift_like(rng, n)
        d1 = np.clip(np.rint(d0 + rng.normal(0, rng.uniform(4.0, 40.0, (n, 1)), (n, 128))), 0, 255).astype(np.float32)
        perm = rng.permutation(n)
        d1p, kp1 = d1[perm], p["pts1"][perm]                                  # the second view lists its keypoints in its own order
        r0, r1 = O.sift_ratio_match(d0, d1p, p["pts0"], kp1, 0.8)
        ref_pts = np.concatenate([r0, r1], 1) if len(r0) else np.full((1, 4), np.nan)
        dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        m = DO.DescriptorRatioMatcher(0.8, dev)([(p["pts0"], d0)], [(kp1, d1p)])
        e = em(m["pts0"], m["pts1"], m["n_corr"], dv(p["K0"][None]), dv(p["K1"][None]), dv(np.array([s], np.int64)))
        torch.cuda.synchronize()
        nn = int(m["n_corr"][0])
        hp = np.concatenate([m["pts0"][0, :nn].cpu().numpy(), m["pts1"][0, :nn].cpu().numpy()], 1)
        if len(r0) >= 5:
            eo = O.emat_solve(r0, r1, p["K0"], p["K1"], 2.0, 0.9999, 1000, 0, s)
            ref = dict(pts=ref_pts, status=int(eo["status"]), R=eo["R"], t=eo["t"].reshape(3), n_inliers=int(eo["n_inl"]), mask=eo["mask"].astype(bool))
        else:
            ref = dict(pts=ref_pts, status=1, R=np.full((3, 3), np.nan), t=np.full(3, np.nan), n_inliers=0)
        r = PR.compare_pair(ref, hp, e["R"][0].cpu().numpy(), e["t"][0].cpu().numpy(), int(e["n_inliers"][0]), int(e["status"][0]),
                            hip_mask=e["mask"][0, :nn].cpu().numpy())
        r["seed"] = int(s)
        if int(e["status"][0]) == 0:
            r["rot_err_vs_truth_deg"] = synth.rot_err_deg(e["R"][0].cpu().numpy(), p["R_gt"])
        recs.append(r)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sg", type=int, default=64)
    ap.add_argument("--loftr", type=int, default=16)
    ap.add_argument("--procrustes", type=int, default=16)
    ap.add_argument("--sift", type=int, default=32)
    ap.add_argument("--hard", type=int, default=2, help="0 easy, 1 moving objects + occluder, 2 = 1 + corrupted depth blocks")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_census.json"))
    a = ap.parse_args()
    res = {"scenes": {0: "easy (3 depth bands)", 1: "hard (moving objects + occluder, images.synthetic_pair(hard=1))",
                      2: "hard=2 (moving objects + occluder + 35-55 % of each depth map wrong by a factor 1.2-1.8, images.synthetic_pair(hard=2))"}[int(a.hard)]}
    for kind, n, chunk in (("sg_pnp", a.sg, 8), ("loftr_emat", a.loftr, 4), ("sg_procrustes", a.procrustes, 8), ("sift_emat", a.sift, 1)):
        if n <= 0:
            continue
        t0 = time.perf_counter()
        recs = census(kind, [5000 + i for i in range(n)], chunk=chunk, hard=int(a.hard))
        s = PR.summarize(recs)
        tr = [r["rot_err_vs_truth_deg"] for r in recs if "rot_err_vs_truth_deg" in r]
        if tr:
            s["median_rot_err_vs_truth_deg"] = round(float(np.median(tr)), 5)
        tt = [r["trans_err_vs_truth_m"] for r in recs if "trans_err_vs_truth_m" in r]
        if tt:
            s["median_trans_err_vs_truth_m"] = round(float(np.median(tt)), 5)
        res[kind] = dict(summary=s, seconds=round(time.perf_counter() - t0, 1), pairs=recs)
        print(kind, json.dumps(s), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
