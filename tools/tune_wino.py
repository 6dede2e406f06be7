"""A/B timing of the Winograd convolution kernel variants (csrc/winograd_conv.hip) on the layer shapes of the two
matchers, with a bit-equality check against variant 1.  python tools/tune_wino.py [--out gpurun_out/tune_wino.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapfree_reloc_amd as mfr  # noqa: E402

SHAPES = [  # name, B, Cin, Cout, H, W, pool
    ("sp.conv1b", 64, 64, 64, 720, 540, 1), ("sp.conv2a", 64, 64, 64, 360, 270, 0), ("sp.conv2b", 64, 64, 64, 360, 270, 1),
    ("sp.conv3a", 64, 64, 128, 180, 135, 0), ("sp.conv3b", 64, 128, 128, 180, 135, 1), ("sp.conv4a", 64, 128, 128, 90, 67, 0), ("sp.conv3b_even", 64, 128, 128, 180, 136, 0),
    ("sp.convPa", 64, 128, 256, 90, 67, 0),
    ("loftr.layer1", 32, 128, 128, 360, 272, 0), ("loftr.layer2", 32, 224, 224, 180, 136, 0), ("loftr.layer3", 32, 256, 256, 90, 68, 0),
    ("loftr.l1out2", 32, 224, 224, 360, 272, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/tune_wino.json")
    ap.add_argument("--variants", default="1,2,4")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--shapes", default="")
    a = ap.parse_args()
    lib = mfr._lib.load(require_gpu=True)
    dev = "cuda"
    variants = [int(v) for v in a.variants.split(",")]
    res = []
    sel = [t for t in SHAPES if not a.shapes or t[0] in a.shapes.split(',')]
    for name, B, Ci, Co, H, W, pool in sel:
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn(B, Ci, H, W, device=dev, generator=g)
        w = torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * 0.05
        b = torch.randn(Co, device=dev, generator=g)
        u = torch.empty(lib.mfr_wino_filter_bytes(Ci, Co) // 4, device=dev)
        mfr._lib.check(lib.mfr_wino_filter_transform(w.data_ptr(), Ci, Co, u.data_ptr(), mfr._lib.stream_ptr()), "ft")
        tiles = ((H + 1) // 2) * ((W + 1) // 2)
        flops = 16 * 2.0 * Ci * Co * tiles * B
        torch.cuda.synchronize()
        ref, row = None, {"shape": name, "B": B, "Cin": Ci, "Cout": Co, "H": H, "W": W, "pool": pool, "gflop": flops / 1e9}
        for v in variants:
            y = torch.empty((B, Co, H // 2, W // 2) if pool else (B, Co, H, W), device=dev)
            call = lambda: mfr._lib.check(lib.mfr_conv3x3_wino_variant(x.data_ptr(), u.data_ptr(), b.data_ptr(), None, B, Ci, Co, H, W, 1, pool, v,
                                                                       y.data_ptr(), mfr._lib.stream_ptr()), "conv")
            for _ in range(2):
                call()
            torch.cuda.synchronize()
            evs = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(p.elapsed_time(q) for p, q in evs)[len(evs) // 2]
            if ref is None:
                ref = y.clone()
            row[f"v{v}_ms"] = round(ms, 4); row[f"v{v}_tflops"] = round(flops / ms / 1e9, 1); row[f"v{v}_equal"] = bool(torch.equal(ref, y))
            row[f"v{v}_maxdiff"] = float((ref - y).abs().max()) if v < 10 else None
        print(json.dumps(row), flush=True)
        res.append(row)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
