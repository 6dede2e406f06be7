"""can the whole SuperGlue+PnP step be captured in a HIP graph? (diagnostic)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapfree_reloc_amd import images as IM
from mapfree_reloc_amd.pipeline import SuperGluePnPPipeline
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sb = IM.synthetic_batch(list(range(B)))
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb.items()}
sb2 = IM.synthetic_batch(list(range(100, 100 + B)))
d2 = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb2.items()}
pipe = SuperGluePnPPipeline(dev)
static = {k: d[k].clone() for k in ("images", "depth0", "K0", "K1", "pair_ids")}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        out = pipe(static["images"], static["depth0"], static["K0"], static["K1"], static["pair_ids"])
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in out.items()}
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = pipe(static["images"], static["depth0"], static["K0"], static["K1"], static["pair_ids"])
torch.cuda.synchronize()
print("captured")
g.replay(); torch.cuda.synchronize()
print("replay equal:", all(torch.equal(gout[k], ref[k]) or (torch.isnan(gout[k]) == torch.isnan(ref[k])).all() for k in ref))
for k in static: static[k].copy_(d2[k])
g.replay(); torch.cuda.synchronize()
e = pipe(d2["images"], d2["depth0"], d2["K0"], d2["K1"], d2["pair_ids"]); torch.cuda.synchronize()
print("batch2 equal to eager:", torch.equal(gout["R"], e["R"]), torch.equal(gout["n_inliers"], e["n_inliers"]))
t0 = time.perf_counter()
for i in range(10):
    for k in static: static[k].copy_((d if i & 1 else d2)[k])
    g.replay()
torch.cuda.synchronize()
print("graph: %.1f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3))
