"""A/B timing of the 3x3 convolution kernels on the SuperPoint / LoFTR layer shapes of the bench batch: exact-fp32 Winograd
(mfr_conv3x3_wino) vs the operand-splitting Winograd kernel in both arithmetics (mfr_conv3x3_wino_bf16x3, mfr_conv3x3_wino_f16x2).
python tools/bench_conv.py [out.json] [images]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd import _lib

lib = _lib.load(require_gpu=True)
dev = "cuda:0"
NIMG = int(sys.argv[2]) if len(sys.argv) > 2 else 64
LAYERS = [("sp.conv1b", 64, 64, 720, 540, 1), ("sp.conv2a", 64, 64, 360, 270, 0), ("sp.conv2b", 64, 64, 360, 270, 1),
          ("sp.conv3a", 64, 128, 180, 135, 0), ("sp.conv3b", 128, 128, 180, 135, 1), ("sp.conv4a", 128, 128, 90, 67, 0),
          ("sp.convPa", 128, 256, 90, 67, 0),
          ("loftr.layer1 128->128 @360x272", 128, 128, 360, 272, 0), ("loftr.layer2 196->196 @180x136", 196, 196, 180, 136, 0),
          ("loftr.layer3 256->256 @90x68", 256, 256, 90, 68, 0), ("loftr.l1out2.0 196->196 @360x272", 196, 196, 360, 272, 0)]
res = {}
for name, ci, co, H, W, pool in LAYERS:
    n = NIMG if not name.startswith("loftr") else max(NIMG // 2, 1)
    x = torch.randn(n, ci, H, W, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) / (3.0 * ci ** 0.5)
    b = torch.randn(co, device=dev)
    y = torch.empty((n, co, H // 2, W // 2) if pool else (n, co, H, W), device=dev)
    u1 = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, dtype=torch.float32, device=dev)
    lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u1), _lib.stream_ptr())
    u3 = torch.empty(lib.mfr_wino_bf16x3_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
    lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u3), _lib.stream_ptr())
    u2 = torch.empty(lib.mfr_wino_f16x2_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
    lib.mfr_wino_f16x2_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u2), _lib.stream_ptr())
    rec = {}
    for tag, fn, u in (("exact_fp32", lib.mfr_conv3x3_wino, u1), ("bf16x3", lib.mfr_conv3x3_wino_bf16x3, u3), ("f16x2", lib.mfr_conv3x3_wino_f16x2, u2),
                       ("exact_fp32_b", lib.mfr_conv3x3_wino, u1), ("bf16x3_b", lib.mfr_conv3x3_wino_bf16x3, u3), ("f16x2_b", lib.mfr_conv3x3_wino_f16x2, u2)):
        for _ in range(2):
            fn(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, n, ci, co, H, W, 1, pool, _lib.ptr(y), _lib.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, n, ci, co, H, W, 1, pool, _lib.ptr(y), _lib.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        rec[tag] = round(e0.elapsed_time(e1) / 5, 4)
    wino_flops = 16 * 2.0 * ci * co * ((H + 1) // 2) * ((W + 1) // 2) * n
    rec["images"] = n
    rec["fp32_equiv_tflops_f16x2"] = round(wino_flops / min(rec["f16x2"], rec["f16x2_b"]) / 1e9, 1)
    rec["fp32_equiv_tflops_bf16x3"] = round(wino_flops / min(rec["bf16x3"], rec["bf16x3_b"]) / 1e9, 1)
    rec["fp32_tflops_exact"] = round(wino_flops / min(rec["exact_fp32"], rec["exact_fp32_b"]) / 1e9, 1)
    rec["f16_mfma_tflops_f16x2"] = round(3 * wino_flops / min(rec["f16x2"], rec["f16x2_b"]) / 1e9, 1)
    rec["bf16_mfma_tflops_bf16x3"] = round(6 * wino_flops / min(rec["bf16x3"], rec["bf16x3_b"]) / 1e9, 1)
    res[name] = rec
    print(name, rec, flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
