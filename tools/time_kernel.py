"""time individual C-ABI stages with HIP events (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd.nets import weights as WT
from mapfree_reloc_amd.nets.superpoint import SuperPointHIP
from mapfree_reloc_amd.nets.superglue import SuperGlueHIP
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
sp = SuperPointHIP(WT.superpoint_state_dict(), dev)
sg = SuperGlueHIP(WT.superglue_state_dict(), dev)
g = torch.Generator().manual_seed(0)
scores = (torch.rand(32, 720, 536, generator=g) ** 3).to(dev)
print("nms 32 images: %.3f ms" % timeit(lambda: sp.nms_candidates(scores)))
qkv = torch.randn(32, 1024, 768, generator=g).to(dev); n = torch.full((32,), 1024, dtype=torch.int32, device=dev)
t = timeit(lambda: sg.attention(qkv, n, False)); print("attention self: %.3f ms -> %.1f TF" % (t, 32 * 4 * 4 * 1024 * 1024 * 64 / t / 1e9))
t = timeit(lambda: sg.attention(qkv, n, True)); print("attention cross: %.3f ms -> %.1f TF" % (t, 32 * 4 * 4 * 1024 * 1024 * 64 / t / 1e9))
S = torch.randn(16, 1024, 1024, generator=g).to(dev); k = torch.rand(16, 1024, 2, generator=g).to(dev)
n16 = torch.full((16,), 1024, dtype=torch.int32, device=dev)
print("sinkhorn+match 16 pairs: %.3f ms" % timeit(lambda: sg.sinkhorn_match(S, n16, n16, k, k)))
