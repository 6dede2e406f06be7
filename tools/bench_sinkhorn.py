import sys, torch
sys.path.insert(0, "/root/repo")
import mapfree_reloc_amd as m
from mapfree_reloc_amd.nets.superglue import SuperGlueHIP
from mapfree_reloc_amd.nets import weights as WT
dev = "cuda:0"
sg = SuperGlueHIP(WT.superglue_state_dict(), dev)
B, K = 32, 1024
g = torch.Generator().manual_seed(1)
S = (torch.randn(B, K, K, generator=g) * 2).to(dev)
n = torch.full((B,), K, dtype=torch.int32, device=dev)
k0 = torch.rand(B, K, 2, device=dev) * 500; k1 = torch.rand(B, K, 2, device=dev) * 500
for variant in (1, 0, 1, 0):
    for _ in range(3): sg.sinkhorn_match(S, n, n, k0, k1, variant=variant)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): sg.sinkhorn_match(S, n, n, k0, k1, variant=variant)
    e1.record(); torch.cuda.synchronize()
    print("variant", variant, round(e0.elapsed_time(e1) / 10, 4), "ms per 32 pairs")
