"""End-to-end parity census: the whole HIP pipeline vs the whole CPU-oracle pipeline on N synthetic
pairs per config -- per pair: match set identical?, pose delta, inlier-count delta.  Writes one JSON (summary +
per-pair records) for profiles/.  Usage: python tools/parity_census.py [--sg 32] [--loftr 8] [--out gpurun_out/census.json]"""
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
from mapfree_reloc_amd.pipeline import LoFTREmatPipeline, SuperGluePnPPipeline  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402


def census(kind, seeds, dev="cuda", chunk=8, threads=16):
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    pipe = SuperGluePnPPipeline(dev) if kind == "sg_pnp" else LoFTREmatPipeline(dev)
    recs = []
    for lo in range(0, len(seeds), chunk):
        ss = seeds[lo:lo + chunk]
        sb = IM.synthetic_batch(ss)
        d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb.items()}
        if kind == "sg_pnp":
            out = pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
        else:
            out = pipe(d["images"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pair_ids"])
        torch.cuda.synchronize()
        o = {k: v.cpu().numpy() for k, v in out.items() if isinstance(v, torch.Tensor)}
        for i, s in enumerate(ss):
            if kind == "sg_pnp":
                ref = PR.sg_pnp_pair(sb["images"][2 * i, 0], sb["images"][2 * i + 1, 0], sb["depth0"][i], sb["K0"][i], sb["K1"][i], s)
            else:
                ref = PR.loftr_emat_pair(sb["images"][2 * i, 0], sb["images"][2 * i + 1, 0], sb["depth0"][i], sb["depth1"][i], sb["K0"][i], sb["K1"][i], s)
            n = int(o["n_corr"][i])
            r = PR.compare_pair(ref, np.concatenate([o["pts0"][i, :n], o["pts1"][i, :n]], 1), o["R"][i], o["t"][i], o["n_inliers"][i], o["status"][i])
            r["seed"] = int(s)
            recs.append(r)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sg", type=int, default=32)
    ap.add_argument("--loftr", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_census.json"))
    a = ap.parse_args()
    res = {}
    for kind, n in (("sg_pnp", a.sg), ("loftr_emat", a.loftr)):
        if n <= 0:
            continue
        t0 = time.perf_counter()
        recs = census(kind, [5000 + i for i in range(n)])
        res[kind] = dict(summary=PR.summarize(recs), seconds=round(time.perf_counter() - t0, 1), pairs=recs)
        print(kind, json.dumps(res[kind]["summary"]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
