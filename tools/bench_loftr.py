"""Secondary measurement (not the driver's bench line): BASELINE configs[2] = LoFTR + E-matrix RANSAC + metric scale
from depth, 540x720 (right-padded to 544 like the reference, quirk Q3), B pairs resident in HBM, 1 GPU.
Usage: python tools/bench_loftr.py [--batch 8] [--steps 6] [--warmup 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapfree_reloc_amd import images as IM  # noqa: E402
from mapfree_reloc_amd.pipeline import LoFTREmatPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    batches = []
    for k in range(2):
        sb = IM.synthetic_batch([100 * k + i for i in range(a.batch)], 720, 540)
        batches.append({key: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for key, v in sb.items()})
    pipe = LoFTREmatPipeline(dev)

    def step(i):
        d = batches[i & 1]
        return pipe(d["images"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pair_ids"])
    for i in range(a.warmup + 2):
        out = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": "image-pairs/sec @ 540x720 (LoFTR + E-mat w/ scale from depth)", "value": round(a.batch * a.steps / el, 2),
                      "unit": "image-pairs/s", "n_gpus": 1, "steps": a.steps, "ms_per_step": round(1e3 * el / a.steps, 2),
                      "config": {"workload": "configs[2]: LoFTR + E-mat + scale, 540x720 (padded 544)", "pairs_per_step": a.batch,
                                 "mean_matches_last_step": float(out["n_corr"].float().mean()),
                                 "pairs_solved_last_step": int((out["status"] == 0).sum())},
                      "dtype": "f32 (matcher) / f64 (solver)", "data": "synthetic scenes, seeded random weights"}))


if __name__ == "__main__":
    main()
