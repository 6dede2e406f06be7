"""A/B timing of the dual-softmax coarse matching variants at the Map-free size (6120 x 6120 per pair): variant 0 = two sweeps
over S (tiled kernels), variant 1 = four sweeps (round 1).  Prints one JSON line.  usage: ab_coarse_match.py [pairs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd.nets import weights as WT
from mapfree_reloc_amd.nets.loftr import LoFTRHIP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
net = LoFTRHIP(WT.loftr_state_dict(), dev)
h, w = 90, 68
L = h * w
g = torch.Generator(device="cuda").manual_seed(0)
f0 = torch.randn(B, L, 256, device=dev, generator=g) * 2.2
f1 = f0[:, torch.randperm(L, device=dev, generator=g)] + 0.3 * torch.randn(B, L, 256, device=dev, generator=g)
S = torch.bmm(f0 / 256.0, f1.transpose(1, 2))
res = {"pairs": B, "L": L, "S_bytes": S.numel() * 4}
for v in (1, 0, 1, 0):
    net.coarse_match(S, (h, w), (h, w), variant=v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = net.coarse_match(S, (h, w), (h, w), variant=v)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res[f"variant{v}_ms"] = round(ms, 4)
    res[f"variant{v}_sweep_equiv_GBs"] = round((2 if v == 0 else 4) * S.numel() * 4 / (ms * 1e-3) / 1e9, 1)
    res[f"variant{v}_matches"] = int(out[3].sum())
print(json.dumps(res))
