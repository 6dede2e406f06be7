"""Time the SuperPoint encoder convolutions (B=32 images, 540x720) through MIOpen under different
PyTorch settings: default immediate mode, cudnn.benchmark (MIOpen find), channels_last.
Usage: python tools/diag_conv.py [default|benchmark|channels_last ...]"""
import sys
import time

import torch
import torch.nn.functional as F

LAYERS = [("conv1b", 64, 64, 540, 720), ("conv2a", 64, 64, 270, 360), ("conv2b", 64, 64, 270, 360),
          ("conv3a", 64, 128, 135, 180), ("conv3b", 128, 128, 135, 180), ("conv4a", 128, 128, 67, 90),
          ("conv4b", 128, 128, 67, 90), ("convPa", 128, 256, 67, 90), ("convDa", 128, 256, 67, 90)]


def run(mode, B=32):
    torch.backends.cudnn.benchmark = (mode == "benchmark")
    dev = torch.device("cuda")
    tot = 0.0
    for name, ci, co, h, w in LAYERS:
        x = torch.randn(B, ci, h, w, device=dev)
        wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        if mode == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last)
            wt = wt.contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            y = F.conv2d(x, wt, None, padding=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = F.conv2d(x, wt, None, padding=1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gf = 2 * 9 * ci * co * h * w * B / 1e9
        tot += ms
        print(f"{mode:14s} {name} {ms:8.3f} ms  {gf / ms:8.1f} TFLOP/s(direct-equivalent x1e-3)", flush=True)
        del x, y
    print(f"{mode:14s} total {tot:.3f} ms", flush=True)


if __name__ == "__main__":
    t = time.time()
    for m in (sys.argv[1:] or ["default", "benchmark", "channels_last"]):
        run(m)
    print("wall", time.time() - t)
