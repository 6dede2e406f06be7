"""per-statement wall time of the batch-1 model plugin (SuperGlue + PnP): where do the milliseconds go on the host?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mapfree_reloc_amd.builder import build_model
from mapfree_reloc_amd.datasets import SyntheticScene, collate_batch1
from tools.bench_plugin import cfg_for

print("torch threads", torch.get_num_threads(), "host cores", os.cpu_count(), flush=True)
sc = SyntheticScene(0, frames=8)
samples = [collate_batch1(sc[i]) for i in range(8)]
model = build_model(cfg_for("SuperGlue", "PNP"))
fm, ps = model.feature_matching, model.pose_solver
for s in samples[:2]:
    model(s)
torch.cuda.synchronize()
T = {}


def tick(name, t0):
    T.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))


for s in samples:
    t_all = time.perf_counter()
    t0 = time.perf_counter(); ims = fm._gray(s); tick("gray+pack (numpy)", t0)
    key = tuple(ims.shape)
    t0 = time.perf_counter(); out = fm._graphs[key](ims); tick("H2D + graph replay (issue)", t0)
    t0 = time.perf_counter(); flat = out.cpu().numpy(); tick("D2H + wait for the GPU", t0)
    n = int(flat[0]); K = (len(flat) - 1) // 4
    k0, k1 = flat[1:1 + 2 * K].reshape(K, 2)[:n].copy(), flat[1 + 2 * K:].reshape(K, 2)[:n].copy()
    t0 = time.perf_counter(); R, t, c = ps.estimate_pose(k0, k1, s); tick("estimate_pose", t0)
    t0 = time.perf_counter()
    as_f32 = lambda a, shape: torch.from_numpy(np.array(a, dtype=np.float32, copy=True).reshape(shape))
    r = as_f32(R, (1, 3, 3)), as_f32(t, (1, 1, 3)); tick("as_f32", t0)
    tick("sum of the statements", t_all)
    t0 = time.perf_counter(); model(s); tick("model(data) as a whole", t0)
for k, v in T.items():
    print(f"{k:32s} median {np.median(v):7.2f} ms   min {min(v):7.2f}   max {max(v):7.2f}")
