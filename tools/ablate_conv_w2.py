"""Timing ablations of the two-workgroups-per-CU bf16x3 Winograd kernel on SuperPoint's conv1b (64 -> 64 channels at 720x540, 64 images,
fused 2x2 max-pool): variant 200 + ABL of mfr_conv3x3_wino_bf16x3_variant (ABL bits: 1 no output transform / stores, 2 no filter
loads, 4 no V production, 8 no patch DMA / LDS patch reads, 64 no MFMAs; results of those builds are WRONG by construction).
python tools/ablate_conv_w2.py [out.json]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mapfree_reloc_amd import _lib

lib = _lib.load(require_gpu=True)
dev = "cuda:0"
n, ci, co, H, W = 64, 64, 64, 720, 540
x = torch.randn(n, ci, H, W, device=dev)
w = torch.randn(co, ci, 3, 3, device=dev) / 24.0
b = torch.randn(co, device=dev)
y = torch.empty(n, co, H // 2, W // 2, device=dev)
u3 = torch.empty(lib.mfr_wino_bf16x3_filter_bytes(ci, co), dtype=torch.uint8, device=dev)
lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u3), _lib.stream_ptr())
res = {}
names = {32: "one wavefront per SIMD (round 3)", 3: "EIGHT wavefronts (p8)", 471: "p8 patches ONLY (DMA + LDS reads + row combos)", 479: "p8 nothing (launch + barriers)", 503: "p8 patch DMA only, nothing else", 487: "p8 LDS reads only, nothing else", 401: "p8 - output transform", 402: "p8 - filter loads", 404: "p8 - V production", 408: "p8 - patch DMA + LDS reads", 416: "p8 - patch DMA only", 432: "p8 - LDS patch reads only", 415: "p8 MFMA only", 464: "p8 no MFMA", 2: "two workgroups per CU", 201: "- output transform", 202: "- filter loads", 204: "- V production",
         208: "- patch DMA + LDS reads", 212: "- V - patches", 214: "- V - patches - filters", 215: "MFMA only", 264: "no MFMA",
         216: "- patch DMA only", 232: "- LDS patch reads only", 301: "TUNE 1: no fences in V production (correct)",
         302: "TUNE 2: row combinations 4 channels at a time (correct)", 303: "TUNE 3: both (correct)"}
for rnd in range(2):
    for v in names:
        for _ in range(2):
            rc = lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u3), _lib.ptr(b), None, n, ci, co, H, W, 1, 1, v, _lib.ptr(y), _lib.stream_ptr())
            assert rc == 0, (v, rc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u3), _lib.ptr(b), None, n, ci, co, H, W, 1, 1, v, _lib.ptr(y), _lib.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        res.setdefault(names[v], []).append(round(e0.elapsed_time(e1) / 5, 3))
for k, v in res.items():
    print(f"{k:40s} {v}")
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
