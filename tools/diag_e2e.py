"""stage-by-stage timing of the fused pipeline (diagnostic; prints with flush)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
t00 = time.time()
def log(*a):
    print(f"[{time.time()-t00:7.2f}s]", *a, flush=True)
import mapfree_reloc_amd as mfr
from mapfree_reloc_amd import images as IM
from mapfree_reloc_amd.pipeline import SuperGluePnPPipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
log("gen data B=", B)
sb = IM.synthetic_batch(list(range(B)))
d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sb.items()}
log("build pipe")
pipe = SuperGluePnPPipeline(dev)
def sync(tag):
    torch.cuda.synchronize(); log(tag)
for rep in range(3):
    x = pipe.sp.encode(d["images"]); sync("sp.encode")
    logits = pipe.sp._conv(pipe.sp._conv(x, "convPa"), "convPb", relu=False); sync("det head")
    scores = pipe.sp.score_map(logits); sync("score_map")
    cand, cnt, _ = pipe.sp.nms_candidates(scores); sync("nms"); log("cand counts", cnt[:4].tolist())
    kpts, sc, n = pipe.sp.select(cand, cnt, scores.shape[2]); sync("select"); log("n", n[:4].tolist())
    spo = pipe.sp(d["images"]); sync("sp full")
    m = pipe.sg(spo, (720, 540), maxN=1024); sync("sg full"); log("n_corr", m["n_corr"][:4].tolist())
    out = pipe.pnp(m["pts0"], m["pts1"], m["n_corr"], d["depth0"], d["K0"], d["K1"], d["pair_ids"]); sync("pnp")
    log("status", out["status"][:8].tolist(), "inl", out["n_inliers"][:8].tolist())
    log("t err", (out["t"][:4] - d["t_gt"][:4]).abs().max().item())
t0 = time.time()
for rep in range(3):
    out = pipe(d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
torch.cuda.synchronize()
log("3 full steps: %.1f ms/step -> %.1f pairs/s" % ((time.time() - t0) / 3 * 1e3, 3 * B / (time.time() - t0)))
