"""Solver-only measurement (SURVEY.md 8d "synthetic correspondences for solver-only benches"; BASELINE configs[0] =
SIFT + E-mat is the reference's CPU-runnable case): B = 32 pairs, N correspondences each (30 % outliers, 1 px noise),
GPU batch time vs the CPU oracle (1 thread, a sample of pairs).  Also times the SIFT descriptor leg (rootSIFT + exact 2-NN
+ ratio test, 2048 x 2048 descriptors per pair).  Prints one JSON line per (N, outlier fraction).
Usage: python tools/bench_solvers.py [--n 1024] [--outliers 0.3] | --sweep   (the 8d grid: N in {256, 1024, 4096} x outliers {0.2, 0.5})"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapfree_reloc_amd import descriptor_ops as D, solver_ops as ops, synth  # noqa: E402
from oracle import oracle_lib as O  # noqa: E402  (checker / CPU baseline only)

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--outliers", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--cpu-pairs", type=int, default=4)
    ap.add_argument("--sweep", action="store_true")
    a = ap.parse_args()
    grid = [(n, o) for n in (256, 1024, 4096) for o in (0.2, 0.5)] if a.sweep else [(a.n, a.outliers)]
    for n, o in grid:
        one(a.batch, n, o, a.cpu_pairs, sift=(not a.sweep) or (n, o) == grid[0])


def one(B, N, outl, cpu_pairs, sift=True):
    bt = synth.make_batch(list(range(B)), [N] * B, maxN=N, outlier_frac=outl, noise_px=1.0, depth_noise=0.002, zero_depth_frac=0.02)
    g = {k: dev(bt[k]) for k in ("pts0", "pts1", "n_corr", "depth0", "depth1", "K0", "K1", "pair_ids")}
    res = {"pairs": B, "n_corr": N, "outlier_frac": outl, "gpu_ms_per_batch": {}, "cpu_ms_per_pair_1thread": {}}
    emat = ops.EssentialBatchSolver(2.0, 0.9999, 0)
    emat_count = ops.EssentialBatchSolver(2.0, 0.9999, 0, score="count")
    scale = ops.ScaleFromDepthBatch(0.1)
    pnp = ops.PnPBatchSolver(1000, 3.0, 0.9999, 0)
    proc = ops.ProcrustesBatchSolver(0.05, 0.999, 0, 4096)
    e = emat(g["pts0"], g["pts1"], g["n_corr"], g["K0"], g["K1"], g["pair_ids"])
    res["gpu_ms_per_batch"]["emat_ransac"] = timed(lambda: emat(g["pts0"], g["pts1"], g["n_corr"], g["K0"], g["K1"], g["pair_ids"]))
    res["gpu_ms_per_batch"]["emat_ransac_count_score"] = timed(lambda: emat_count(g["pts0"], g["pts1"], g["n_corr"], g["K0"], g["K1"], g["pair_ids"]))
    d = emat(g["pts0"], g["pts1"], g["n_corr"], g["K0"], g["K1"], g["pair_ids"], diagnostics=True)
    res["emat_iters_run_mean"] = float(d["iters_run"].double().mean()); res["emat_lo_runs_mean"] = float(d["lo_runs"].double().mean())
    rot = [synth.rot_err_deg(d["R"][b].cpu().numpy(), bt["R_gt"][b]) for b in range(B) if int(d["status"][b]) == 0]
    res["emat_rot_err_deg_median"] = float(np.median(rot)) if rot else None
    res["gpu_ms_per_batch"]["scale_from_depth"] = timed(lambda: scale(g["pts0"], g["pts1"], e["mask"], g["n_corr"], g["depth0"], g["depth1"],
                                                                       g["K0"], g["K1"], e["R"], e["t"], e["status"]))
    res["gpu_ms_per_batch"]["pnp_ransac"] = timed(lambda: pnp(g["pts0"], g["pts1"], g["n_corr"], g["depth0"], g["K0"], g["K1"], g["pair_ids"]))
    res["gpu_ms_per_batch"]["procrustes_ransac"] = timed(lambda: proc(g["pts0"], g["pts1"], g["n_corr"], g["depth0"], g["depth1"], g["K0"], g["K1"],
                                                                      g["pair_ids"]))
    P = cpu_pairs
    if sift:
        sift_leg_bench(res, B)
    # CPU oracle, one thread, a few pairs
    t0 = time.perf_counter()
    for b in range(P):
        O.emat_solve(bt["pts0"][b], bt["pts1"][b], bt["K0"][b], bt["K1"][b], 2.0, 0.9999, 1000, 0, int(bt["pair_ids"][b]))
    res["cpu_ms_per_pair_1thread"]["emat_ransac"] = 1e3 * (time.perf_counter() - t0) / P
    t0 = time.perf_counter()
    for b in range(P):
        O.pnp_solve(bt["pts0"][b], bt["pts1"][b], bt["depth0"][b], bt["K0"][b], bt["K1"][b], 1000, 3.0, 0.9999, 0, int(bt["pair_ids"][b]))
    res["cpu_ms_per_pair_1thread"]["pnp_ransac"] = 1e3 * (time.perf_counter() - t0) / P
    t0 = time.perf_counter()
    for b in range(P):
        O.procrustes_solve(bt["pts0"][b], bt["pts1"][b], bt["depth0"][b], bt["depth1"][b], bt["K0"][b], bt["K1"][b], 0.05, 0.999, 4096, 0,
                           int(bt["pair_ids"][b]))
    res["cpu_ms_per_pair_1thread"]["procrustes_ransac"] = 1e3 * (time.perf_counter() - t0) / P
    res["pairs_per_s_gpu"] = {k: round(1e3 * B / v, 1) for k, v in res["gpu_ms_per_batch"].items()}
    res["pairs_per_s_cpu_1thread"] = {k: round(1e3 / v, 2) for k, v in res["cpu_ms_per_pair_1thread"].items()}
    print(json.dumps(res), flush=True)


def sift_leg_bench(res, B):
    # SIFT descriptor leg: 2048 x 2048 rootSIFT descriptors per pair
    rng = np.random.default_rng(0)
    des = torch.from_numpy(rng.integers(0, 120, (2 * B, 2048, 128)).astype(np.float32)).to(DEV)
    kp = torch.rand(2 * B, 2048, 2, device=DEV) * 500
    n = torch.full((B,), 2048, dtype=torch.int32, device=DEV)

    def sift_leg():
        r, q = D.rootsift(des)
        return D.ratio_match(r[0::2].contiguous(), r[1::2].contiguous(), q[0::2].contiguous(), q[1::2].contiguous(),
                             kp[0::2].contiguous(), kp[1::2].contiguous(), n, n, 0.8)
    res["gpu_ms_per_batch"]["sift_rootsift_2nn_ratio_2048x2048"] = timed(sift_leg)
    d0 = des[0].cpu().numpy(); d1 = des[1].cpu().numpy()
    t0 = time.perf_counter()
    O.sift_ratio_match(d0, d1, kp[0].cpu().numpy(), kp[1].cpu().numpy(), 0.8)
    res["cpu_ms_per_pair_1thread"]["sift_rootsift_2nn_ratio_2048x2048"] = 1e3 * (time.perf_counter() - t0)


if __name__ == "__main__":
    main()
