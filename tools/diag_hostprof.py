import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapfree_reloc_amd import images as IM
from mapfree_reloc_amd.pipeline import SuperGluePnPPipeline
dev = torch.device("cuda:0")
B = 16
sb = IM.synthetic_batch(list(range(B)))
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb.items()}
pipe = SuperGluePnPPipeline(dev)
for _ in range(3):
    pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
torch.cuda.synchronize()
def run():
    for i in range(5):
        t0 = time.perf_counter()
        pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
        print("issue ms", 1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
