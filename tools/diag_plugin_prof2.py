import cProfile, pstats, io, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapfree_reloc_amd.datasets import SyntheticScene, collate_batch1
from mapfree_reloc_amd.builder import build_model
from tools.bench_plugin import cfg_for
sc = SyntheticScene(0, frames=6)
samples = [collate_batch1(sc[i]) for i in range(6)]
model = build_model(cfg_for("SuperGlue", "PNP"))
for s in samples[:2]: model(s)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
for s in samples: model(s)
dt = time.perf_counter() - t0; pr.disable()
print("ms/pair", 1e3 * dt / 6)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
