"""One isolated kernel, launched a few times, for `rocprofv3 --pmc` passes (tools/pmc_kernel.sh): python tools/pmc_driver.py <what> [split]
what: conv1ab (SuperPoint conv1a fused into conv1b, 64 images 720x540), gemm (the 512 -> 512 + ReLU layer at M = 65536), gemm_qkv (256 -> 768), conv1b (64 -> 64 channels, 64 images 720x540, pooled), attention (64 images x 4 heads x 1024),
      loftr_l1out2 (196 -> 196 at 360x272, 32 images), loftr_gemm (256 -> 256 at M = 195840), sinkhorn (32 pairs x 1024^2, 20 sweeps)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mapfree_reloc_amd as m  # noqa: F401
from mapfree_reloc_amd import options

what = sys.argv[1]
if len(sys.argv) > 2:
    options.set("SPLIT", sys.argv[2])
dev = "cuda:0"
if what in ("gemm", "gemm_qkv", "loftr_gemm"):
    from mapfree_reloc_amd.nets.linear import SplitLinear
    M, K, N, relu = {"gemm": (65536, 512, 512, True), "gemm_qkv": (65536, 256, 768, False), "loftr_gemm": (195840, 256, 256, False)}[what]
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    lin = SplitLinear(w, b)
    fn = lambda: lin(x, out=y, relu=relu)
elif what in ("conv1b", "loftr_l1out2"):
    from mapfree_reloc_amd.nets.conv import WinoConv3x3
    B, ci, co, H, W, pool, act = {"conv1b": (64, 64, 64, 720, 540, True, 1), "loftr_l1out2": (32, 196, 196, 360, 272, False, 2)}[what]
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) / (3.0 * ci ** 0.5); b = torch.randn(co, device=dev)
    options.set("CONV_KERNEL", "split")
    cv = WinoConv3x3(w, b)
    fn = lambda: cv(x, act=act, pool=pool)
elif what in ("dconv_l1", "dconv_conv2a", "dconv_l1out2"):      # the direct halo-staged kernel (csrc/conv_direct.hip) on LoFTR layer1 (128 -> 128, residual), SuperPoint conv2a, LoFTR l1out2.0
    from mapfree_reloc_amd.nets.conv import DirectConv3x3
    B, ci, co, H, W, act, res = {"dconv_l1": (32, 128, 128, 272, 360, 1, True), "dconv_conv2a": (64, 64, 64, 270, 360, 1, False), "dconv_l1out2": (32, 196, 196, 272, 360, 2, False)}[what]
    x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) / (3.0 * ci ** 0.5); b = torch.randn(co, device=dev)
    r = torch.randn(B, co, H, W, device=dev) if res else None
    cv = DirectConv3x3(w, b)
    fn = lambda: cv(x, act=act, residual=r)
elif what == "conv1ab":
    from mapfree_reloc_amd.nets.superpoint import SuperPointHIP
    from mapfree_reloc_amd.nets import weights as WT
    sp = SuperPointHIP(WT.superpoint_state_dict(), dev)
    img = torch.rand(64, 1, 720, 540, device=dev)
    fn = lambda: sp._conv1ab(img)
elif what == "attention":
    from mapfree_reloc_amd.nets.superglue import SuperGlueHIP
    from mapfree_reloc_amd.nets import weights as WT
    sg = SuperGlueHIP(WT.superglue_state_dict(), dev)
    qkv = torch.randn(64, 1024, 768, device=dev); n = torch.full((64,), 1024, dtype=torch.int32, device=dev); out = torch.empty(64, 1024, 256, device=dev)
    fn = lambda: sg.attention(qkv, n, False, out=out)
elif what == "sinkhorn":        # 32 pairs x 1024 x 1024 scores, 20 iterations (the bench step's shape at full keypoint counts)
    from mapfree_reloc_amd.nets.superglue import SuperGlueHIP
    from mapfree_reloc_amd.nets import weights as WT
    sg = SuperGlueHIP(WT.superglue_state_dict(), dev)
    g = torch.Generator().manual_seed(1)
    S = (torch.randn(32, 1024, 1024, generator=g) * 2).to(dev)
    n = torch.full((32,), 1024, dtype=torch.int32, device=dev)
    k0 = torch.rand(32, 1024, 2, device=dev) * 500; k1 = torch.rand(32, 1024, 2, device=dev) * 500
    fn = lambda: sg.sinkhorn_match(S, n, n, k0, k1)
else:
    raise SystemExit(f"unknown {what}")
for _ in range(4):
    fn()
torch.cuda.synchronize()
