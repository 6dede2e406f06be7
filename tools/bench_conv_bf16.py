"""A/B of the decoder's 3x3 convolutions (resunet.py:112-128 shapes, batch 10 images = one view of TRAINING.BATCH_SIZE 10 pairs):
regression/conv_bf16.py (implicit GEMM on the bf16 matrix cores, csrc/conv_gemm_bf16.hip) vs torch / MIOpen under bf16 autocast;
forward and forward+backward, plus the bare kernel launches (no layout copies) to separate kernel time from host-side glue.
python tools/bench_conv_bf16.py [out.json] [images per call, default 10]"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd.regression import conv_bf16 as CB  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = []
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for name, B, C, N, H, W in (("upconv4/iconv4", NB, 1024, 512, 46, 34), ("upconv3/iconv3", NB, 512, 256, 92, 68)):
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().requires_grad_()
    w = (torch.randn(N, C, 3, 3, device="cuda") * 0.02).requires_grad_()
    b = torch.randn(N, device="cuda").requires_grad_()
    gy = torch.randn(B, N, H, W, device="cuda").bfloat16()
    flops = 2.0 * 9 * C * N * B * H * W

    def lib_fwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return F.conv2d(x, w, b, padding=1)

    def lib_fb():
        y = lib_fwd(); y.backward(gy); x.grad = None; w.grad = None; b.grad = None

    def hip_fwd():
        return CB.conv3x3_bf16(x, w, b)

    def hip_fb():
        y = hip_fwd(); y.backward(gy); x.grad = None; w.grad = None; b.grad = None

    wmat = w.detach().permute(0, 2, 3, 1).reshape(N, 9 * C).bfloat16().contiguous()
    r = dict(layer=name, shape=[B, C, N, H, W], gflop_fwd=round(flops / 1e9, 1))
    for k, fn, mult in (("miopen_fwd", lib_fwd, 1), ("hip_fwd", hip_fwd, 1), ("miopen_fwd_bwd", lib_fb, 3), ("hip_fwd_bwd", hip_fb, 3)):
        ms = timeit(fn)
        r[k] = dict(ms=round(ms, 3), tflops=round(mult * flops / ms / 1e9, 1))
    # the stages of the hip path on their own
    xd, gyd = x.detach(), gy
    r["hip_stage_fwd_haloed (layout copy + kernel)"] = round(timeit(lambda: CB._conv_haloed(xd, wmat, None)), 3)
    r["hip_stage_wgrad (3 shifted copies + kernel + split sum)"] = round(timeit(lambda: CB._wgrad(xd, gyd)), 3)
    r["weight_pack"] = round(timeit(lambda: w.detach().permute(0, 2, 3, 1).reshape(N, 9 * C).bfloat16().contiguous()), 3)
    # bare kernel: forward product only, operands prepared once
    Hp, Wp = H + 2, W + 1
    Mp = B * Hp * Wp
    G = (Wp + 1 + 7) // 8 * 8
    xp = torch.zeros((G + Mp + G) * C, dtype=torch.bfloat16, device="cuda")
    r["workgroups_fwd"] = ((Mp + 255) // 256) * ((N + 127) // 128)
    out = torch.empty(Mp, N, dtype=torch.bfloat16, device="cuda")
    for order in ("tap_outer", "tap_inner"):
        segA, segB, Lk = CB.tap_tables(C, Wp, "cuda", order)
        ms = timeit(lambda: CB.seg_gemm(xp, G * C, C, segA, wmat, 0, 9 * C, segB, Lk, 9 * C // 32, 9 * C // 32, None, out, N, Mp, N))
        r["bare_kernel_fwd_" + order] = dict(ms=round(ms, 3), tflops=round(2.0 * 9 * C * N * Mp / ms / 1e9, 1), rows=Mp)
    print(json.dumps(r), flush=True)
    res.append(r)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
