"""Winograd/MFMA 3x3 convolution (csrc/winograd_conv.hip): correctness against a float64 CPU convolution on
small ragged shapes and against the library convolution at the SuperPoint layer shapes, plus per-layer
timing (B=32 images, 540x720).  Usage: python tools/diag_wino.py [--no-time]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mapfree_reloc_amd as m  # noqa: E402
from mapfree_reloc_amd import _lib  # noqa: E402

# Map-free frames are 540 wide x 720 tall
LAYERS = [("conv1b", 64, 64, 720, 540, 1), ("conv2a", 64, 64, 360, 270, 0), ("conv2b", 64, 64, 360, 270, 1),
          ("conv3a", 64, 128, 180, 135, 0), ("conv3b", 128, 128, 180, 135, 1), ("conv4a", 128, 128, 90, 67, 0),
          ("conv4b", 128, 128, 90, 67, 0), ("convPa", 128, 256, 90, 67, 0), ("convDa", 128, 256, 90, 67, 0)]


def wino(x, w, b, relu, pool):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    u = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "filter")
    y = torch.empty((B, co, H // 2, W // 2) if pool else (B, co, H, W), dtype=torch.float32, device=x.device)

    def run():
        _lib.check(lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None, None, B, ci, co, H, W,
                                        int(relu), int(pool), _lib.ptr(y), _lib.stream_ptr()), "conv")
    run()
    return y, run


def ref64(x, w, b, relu, pool):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)
    if relu:
        y = y.relu()
    if pool:
        y = F.max_pool2d(y, 2, 2)
    return y


def main():
    dev = torch.device("cuda")
    torch.manual_seed(0)
    ok = True
    for (B, ci, co, H, W, relu, pool, bias) in [(1, 8, 32, 8, 32, 0, 0, 0), (2, 8, 32, 11, 38, 1, 0, 1), (1, 64, 64, 17, 45, 1, 1, 1),
                                                 (3, 24, 96, 9, 33, 0, 1, 1), (1, 128, 256, 67, 90, 1, 0, 1), (2, 64, 128, 135, 180, 1, 1, 1),
                                                 (1, 8, 32, 2, 2, 1, 1, 1), (1, 8, 32, 1, 1, 0, 0, 1), (1, 16, 32, 12, 31, 1, 1, 1)]:
        x = torch.randn(B, ci, H, W, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) * (1.0 / (3.0 * ci ** 0.5))
        b = torch.randn(co, device=dev) if bias else None
        y, _ = wino(x, w, b, relu, pool)
        r = ref64(x, w, b, relu, pool)
        err = (y.double().cpu() - r).abs().max().item()
        lib_err = float("nan")
        if not pool:
            yl = F.conv2d(x, w, b, padding=1)
            yl = yl.relu() if relu else yl
            lib_err = (yl.double().cpu() - r).abs().max().item()
        good = err < 2e-5
        ok &= good
        print(f"shape B{B} {ci}->{co} {H}x{W} relu{relu} pool{pool} bias{bias}: max|err| {err:.3e} (library conv {lib_err:.3e}) "
              f"{'OK' if good else 'FAIL'}", flush=True)
    if "--no-time" in sys.argv:
        return 0 if ok else 1
    import os
    for nblk in ("2", "4"):
        os.environ["MFR_WINO_NBLK"] = nblk
        tot_w = 0.0
        for name, ci, co, H, W, pool in LAYERS:
            x = torch.randn(32, ci, H, W, device=dev)
            w = torch.randn(co, ci, 3, 3, device=dev) * (1.0 / (3.0 * ci ** 0.5))
            b = torch.randn(co, device=dev)
            y, run = wino(x, w, b, 1, pool)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            gf = 2 * 9 * ci * co * H * W * 32 / 1e9
            tot_w += ms
            print(f"NBLK<={nblk} {name}: {ms:7.3f} ms ({gf / 2.25 / ms:6.1f} TF on the MFMA pipe)", flush=True)
            del x, y
        print(f"NBLK<={nblk} total {tot_w:.3f} ms", flush=True)
    os.environ.pop("MFR_WINO_NBLK")
    if "--no-lib" in sys.argv:
        return 0 if ok else 1
    tot_w = tot_l = 0.0
    for name, ci, co, H, W, pool in LAYERS:
        B = 32
        x = torch.randn(B, ci, H, W, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) * (1.0 / (3.0 * ci ** 0.5))
        b = torch.randn(co, device=dev)
        y, run = wino(x, w, b, 1, pool)
        yl = F.conv2d(x, w, b, padding=1).relu()
        yl = F.max_pool2d(yl, 2, 2) if pool else yl
        err = (y - yl).abs().max().item()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        e0.record()
        for _ in range(5):
            F.conv2d(x, w, None, padding=1)
        e1.record()
        torch.cuda.synchronize()
        ms_l = e0.elapsed_time(e1) / 5
        gf = 2 * 9 * ci * co * H * W * B / 1e9
        tot_w += ms; tot_l += ms_l
        print(f"{name}: wino {ms:7.3f} ms ({gf / ms:6.1f} TF direct-equiv, {gf / 2.25 / ms:6.1f} TF on the MFMA pipe) | "
              f"library conv only {ms_l:7.3f} ms | max|diff| vs library {err:.2e}", flush=True)
        del x, y, yl
    print(f"total: wino {tot_w:.3f} ms, library conv (without epilogues) {tot_l:.3f} ms")
    return 0 if ok else 1


if __name__ == "__main__":
    t = time.time()
    rc = main()
    print("wall", time.time() - t)
    sys.exit(rc)
