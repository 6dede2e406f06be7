"""Per-pair plugin path (the reference's real usage: build_model(cfg)(data) at batch 1, submission.py:33-58) timed on the GPU
next to the fused batched path.  python tools/bench_plugin.py [--pairs 40] [--out gpurun_out/bench_plugin.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd import synth  # noqa: E402
from mapfree_reloc_amd.builder import build_model  # noqa: E402
from mapfree_reloc_amd.config import get_cfg_defaults  # noqa: E402
from mapfree_reloc_amd.datasets import SyntheticScene, collate_batch1  # noqa: E402
from mapfree_reloc_amd.matching import pose_solver as PS  # noqa: E402


def cfg_for(matcher, solver):
    cfg = get_cfg_defaults()
    cfg.MODEL, cfg.FEATURE_MATCHING, cfg.POSE_SOLVER = "FeatureMatching", matcher, solver
    cfg.PNP.RANSAC_ITER, cfg.PNP.REPROJECTION_INLIER_THRESHOLD, cfg.PNP.CONFIDENCE = 1000, 3, 0.9999
    cfg.EMAT_RANSAC.PIX_THRESHOLD, cfg.EMAT_RANSAC.SCALE_THRESHOLD, cfg.EMAT_RANSAC.CONFIDENCE = 2.0, 0.1, 0.9999
    cfg.PROCRUSTES.MAX_CORR_DIST = 0.05
    cfg.ALLOW_SYNTHETIC_WEIGHTS = True
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=40)
    ap.add_argument("--out", default="gpurun_out/bench_plugin.json")
    ap.add_argument("--hip-opt", action="append", default=[], metavar="NAME=VALUE", help="declared kernel-selection option (options.py), e.g. CONV_KERNEL=split")
    a = ap.parse_args()
    from mapfree_reloc_amd import options
    for kv in a.hip_opt:
        k, v = kv.split("=", 1)
        options.set(k, v)
    res = {}
    # solver plugins alone on synthetic correspondences (1024 per pair, 30 % outliers)
    prs = [synth.make_pair(100 + i, 1024, outlier_frac=0.3) for i in range(a.pairs)]
    datas = [{"depth0": torch.from_numpy(p["depth0"])[None], "depth1": torch.from_numpy(p["depth1"])[None],
              "K_color0": torch.from_numpy(p["K0"])[None], "K_color1": torch.from_numpy(p["K1"])[None], "pair_id": torch.tensor([i])}
             for i, p in enumerate(prs)]
    for name, cls in (("PNP", PS.PnPSolver), ("EssentialMatrixMetric", PS.EssentialMatrixMetricSolver), ("Procrustes", PS.ProcrustesSolver)):
        solver = cls(cfg_for("Precomputed", name))
        for p, d in zip(prs[:3], datas[:3]):
            solver.estimate_pose(p["pts0"], p["pts1"], d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = 0
        for p, d in zip(prs, datas):
            R, t, n = solver.estimate_pose(p["pts0"], p["pts1"], d)
            ok += int(n > 0)
        dt = time.perf_counter() - t0
        res[f"solver_plugin:{name}"] = {"ms_per_pair": round(1e3 * dt / a.pairs, 3), "pairs_per_s": round(a.pairs / dt, 1), "solved": ok}
        print(name, res[f"solver_plugin:{name}"], flush=True)
    # whole model plugin with the online matchers
    sc = SyntheticScene(0, frames=min(a.pairs, 12))
    samples = [collate_batch1(sc[i]) for i in range(len(sc))]
    for matcher, solver in (("SuperGlue", "PNP"), ("LoFTR", "EssentialMatrixMetric")):
        model = build_model(cfg_for(matcher, solver))
        for s in samples[:2]:
            model(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in samples:
            model(s)
        dt = time.perf_counter() - t0
        res[f"model_plugin:{matcher}+{solver}"] = {"ms_per_pair": round(1e3 * dt / len(samples), 2), "pairs_per_s": round(len(samples) / dt, 1)}
        print(matcher, solver, res[f"model_plugin:{matcher}+{solver}"], flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
