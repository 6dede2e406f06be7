"""A/B timing of the SuperGlue linear-layer shapes: library fp32 GEMM (torch / hipBLASLt) vs csrc/gemm_split.hip in both arithmetics"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd.nets.linear import SplitLinear
dev = "cuda:0"
M = 65536
res = {}
for name, K, N, relu, acc in (("qkv 256->768", 256, 768, False, False), ("mlp1 512->512 relu", 512, 512, True, False), ("mlp2 512->256 +=", 512, 256, False, True),
                              ("loftr 256->256 (M=195840)", 256, 256, False, False)):
    m = 195840 if "loftr" in name else M
    x = torch.randn(m, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    y = torch.randn(m, N, device=dev)
    lin3, lin2 = SplitLinear(w, b, split="bf16x3"), SplitLinear(w, b, split="f16x2")
    wt = w.t().contiguous()
    def lib():
        if acc: y.addmm_(x, wt)
        elif relu: torch._addmm_activation(b, x, wt)
        else: torch.addmm(b, x, wt)
    rec = {}
    for tag, fn in (("library_fp32", lib), ("bf16x3", lambda: lin3(x, out=y, relu=relu, accumulate=acc)), ("f16x2", lambda: lin2(x, out=y, relu=relu, accumulate=acc)),
                    ("f16x2_flag8", lambda: lin2(x, out=y, relu=relu, accumulate=acc, kernel_flag=8)),
                    ("library_fp32_b", lib), ("bf16x3_b", lambda: lin3(x, out=y, relu=relu, accumulate=acc)), ("f16x2_b", lambda: lin2(x, out=y, relu=relu, accumulate=acc))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        rec[tag] = round(e0.elapsed_time(e1) / 10, 4)
    fl = 2.0 * m * K * N
    rec["fp32_equiv_tflops_f16x2"] = round(fl / min(rec["f16x2"], rec["f16x2_b"]) / 1e9, 1)
    rec["fp32_equiv_tflops_bf16x3"] = round(fl / min(rec["bf16x3"], rec["bf16x3_b"]) / 1e9, 1)
    rec["tflops_library"] = round(fl / min(rec["library_fp32"], rec["library_fp32_b"]) / 1e9, 1)
    res[name] = rec
    print(name, rec, flush=True)
if len(sys.argv) > 1: json.dump(res, open(sys.argv[1], "w"), indent=1)
