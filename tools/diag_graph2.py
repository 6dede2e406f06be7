"""bisect: batch-1 SuperGlueMatching graph replay standalone"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapfree_reloc_amd.datasets import SyntheticScene, collate_batch1
from mapfree_reloc_amd.matching.feature_matching import SuperGlueMatching
from tools.bench_plugin import cfg_for
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
cfg = cfg_for("SuperGlue", "PNP")
cfg.HIP.GRAPH_BATCH1 = mode != "eager"
sc = SyntheticScene(0, frames=4)
samples = [collate_batch1(sc[i]) for i in range(4)]
m = SuperGlueMatching(cfg)
print("built", flush=True)
for i, s in enumerate(samples):
    a, b = m.get_correspondences(s)
    print("pair", i, len(a), flush=True)
t0 = time.perf_counter()
for s in samples:
    m.get_correspondences(s)
print("ms/pair", 1e3 * (time.perf_counter() - t0) / 4)
