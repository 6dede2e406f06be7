"""A/B record (round 6): the direct f16x2 implicit-GEMM convolution WITH an LDS-staged halo tile (csrc/conv_direct.hip,
mfr_conv3x3_direct_f16x2: every input element split once per workgroup, nine taps as shifted LDS reads) against the split-Winograd kernel on the 3x3
layers of the two backbones at the bench batches (64 SuperPoint images, 32 LoFTR images), same inputs.  Per layer: ms of both, executed f16 TFLOP/s
and the fraction of the dense f16 peak (three partial products), the algorithmic fraction (direct multiply-adds / time / peak).
python tools/ab_direct_halo.py [out.json] [layer-name-substring]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import mapfree_reloc_amd  # noqa: F401
from mapfree_reloc_amd import options
from mapfree_reloc_amd.nets.conv import DirectConv3x3, WinoConv3x3

dev = "cuda:0"
PEAK = 2500.0
# name, images, Cin, Cout, H, W, pool, residual
LAYERS = [("sp.conv1b 64->64 @540x720 pool", 64, 64, 64, 540, 720, 1, 0), ("sp.conv2a 64->64 @270x360", 64, 64, 64, 270, 360, 0, 0),
          ("sp.conv2b 64->64 @270x360 pool", 64, 64, 64, 270, 360, 1, 0), ("sp.conv3a 64->128 @135x180", 64, 64, 128, 135, 180, 0, 0),
          ("sp.conv3b 128->128 @135x180 pool", 64, 128, 128, 135, 180, 1, 0), ("sp.conv4a 128->128 @67x90", 64, 128, 128, 67, 90, 0, 0),
          ("sp.convPa 128->256 @67x90", 64, 128, 256, 67, 90, 0, 0),
          ("loftr.layer1 128->128 @272x360 res", 32, 128, 128, 272, 360, 0, 1), ("loftr.layer2 196->196 @136x180 res", 32, 196, 196, 136, 180, 0, 1),
          ("loftr.layer3 256->256 @68x90 res", 32, 256, 256, 68, 90, 0, 1), ("loftr.l2out2.0 256->256 @136x180", 32, 256, 256, 136, 180, 0, 0),
          ("loftr.l2out2.1 256->196 @136x180", 32, 256, 196, 136, 180, 0, 0), ("loftr.l1out2.0 196->196 @272x360", 32, 196, 196, 272, 360, 0, 0),
          ("loftr.l1out2.1 196->128 @272x360", 32, 196, 128, 272, 360, 0, 0)]


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
options.set("CONV_KERNEL", "split")
sel = sys.argv[2] if len(sys.argv) > 2 else ""
for name, n, ci, co, H, W, pool, has_res in LAYERS:
    if sel not in name:
        continue
    g = torch.Generator().manual_seed(ci + H)
    x = torch.randn(n, ci, H, W, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    r = torch.randn(n, co, H, W, generator=g).to(dev) if has_res else None
    wino, direct = WinoConv3x3(w, b), DirectConv3x3(w, b)
    yw, yd = wino(x, act=1, pool=bool(pool), residual=r), direct(x, act=1, pool=bool(pool), residual=r)
    want = F.conv2d(x[:1].double(), w.double(), b.double(), padding=1)
    if r is not None:
        want = want + r[:1].double()
    want = want.relu()
    if pool:
        want = F.max_pool2d(want, 2, 2)
    t_w = timed(lambda: wino(x, act=1, pool=bool(pool), residual=r))
    t_d = timed(lambda: direct(x, act=1, pool=bool(pool), residual=r))
    flops = 2.0 * 9 * ci * co * H * W * n
    res[name] = dict(images=n, winograd_f16x2_ms=round(t_w, 4), direct_halo_f16x2_ms=round(t_d, 4), speedup=round(t_w / t_d, 3),
                     max_err_vs_f64=dict(winograd=float((yw[:1].double() - want).abs().max()), direct=float((yd[:1].double() - want).abs().max())),
                     direct=dict(executed_tflops=round(3 * flops / t_d / 1e9, 1), mfma_pipe_frac=round(3 * flops / t_d / 1e9 / PEAK, 4),
                                 algorithmic_frac=round(flops / t_d / 1e9 / PEAK, 4)),
                     winograd=dict(algorithmic_frac=round(flops / t_w / 1e9 / PEAK, 4)))
    print(name, json.dumps(res[name]), flush=True)
    del x, w, b, r, wino, direct, yw, yd
options.reset()
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(sys.argv[1]) or ".", exist_ok=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
