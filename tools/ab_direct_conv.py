"""A/B record for VERDICT r5 item 3: a DIRECT f16x2 implicit-GEMM convolution (no Winograd transform) against the split-Winograd kernels on the
64-channel full-resolution SuperPoint layers (conv1b, conv2a, conv2b) and one 128-channel layer, same boxes, same inputs.

The direct kernel measured here is the library's own implicit-GEMM kernel (csrc/gemm_split.hip conv_igemm_f16x2_kernel, mfr_conv_igemm_f16x2: K = 9 taps x
Cin, operands split at staging PER TAP -- it has no LDS halo tile, so every input element is gathered and split nine times).  It bounds from above
what a halo-staged direct kernel would take: the MFMA count of a direct convolution is 2.25 x Winograd F(2x2,3x3)'s, the staging work of the halo
version 1/9 of this kernel's.  Reported per layer: ms, executed f16 TFLOP/s (3 partial products), the MFMA time at 100 % of the dense f16 peak,
and the same for the Winograd kernel.   python tools/ab_direct_conv.py [out.json] [images]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import mapfree_reloc_amd  # noqa: F401
from mapfree_reloc_amd import options
from mapfree_reloc_amd.nets.conv import IgemmConv, WinoConv3x3

dev = "cuda:0"
NIMG = int(sys.argv[2]) if len(sys.argv) > 2 else 64
PEAK = 2500.0
LAYERS = [("sp.conv1b 64->64 @720x540", 64, 64, 720, 540), ("sp.conv2a 64->64 @360x270", 64, 64, 360, 270), ("sp.conv3b 128->128 @180x135", 128, 128, 180, 135)]


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
for name, ci, co, H, W in LAYERS:
    g = torch.Generator().manual_seed(ci + H)
    x = torch.randn(NIMG, ci, H, W, generator=g).to(dev)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    wino, direct = WinoConv3x3(w, b), IgemmConv(w, b, 1)
    yw, yd = wino(x, act=1), direct(x, relu=True)
    want = F.conv2d(x[:2].double(), w.double(), b.double(), padding=1).relu()
    t_w, t_d = timed(lambda: wino(x, act=1)), timed(lambda: direct(x, relu=True))
    direct_flops = 2.0 * 9 * ci * co * H * W * NIMG
    wino_flops = 16 * 2.0 * ci * co * ((H + 1) // 2) * ((W + 1) // 2) * NIMG
    res[name] = dict(images=NIMG, winograd_f16x2_ms=round(t_w, 4), direct_igemm_f16x2_ms=round(t_d, 4),
                     max_err_vs_f64=dict(winograd=float((yw[:2].double() - want).abs().max()), direct=float((yd[:2].double() - want).abs().max())),
                     direct=dict(executed_tflops=round(3 * direct_flops / t_d / 1e9, 1), mfma_pipe_frac=round(3 * direct_flops / t_d / 1e9 / PEAK, 4),
                                 ms_at_100pct_of_the_f16_peak=round(3 * direct_flops / PEAK / 1e9, 3)),
                     winograd=dict(executed_tflops=round(3 * wino_flops / t_w / 1e9, 1), mfma_pipe_frac=round(3 * wino_flops / t_w / 1e9 / PEAK, 4),
                                   ms_at_100pct_of_the_f16_peak=round(3 * wino_flops / PEAK / 1e9, 3)),
                     algorithmic_frac_of_peak=dict(winograd=round(direct_flops / t_w / 1e9 / PEAK, 4), direct=round(direct_flops / t_d / 1e9 / PEAK, 4)),
                     direct_pipe_busy_needed_to_match_winograd=round(3 * direct_flops / PEAK / 1e9 / t_w, 3))
    print(name, json.dumps(res[name]), flush=True)
options.reset()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
