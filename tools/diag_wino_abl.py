"""Tuning aid: time the Winograd conv (conv1b / conv3b shapes) under the MFR_WINO_ABL ablation variants."""
import os
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mapfree_reloc_amd as m  # noqa: E402,F401
from mapfree_reloc_amd import _lib  # noqa: E402


def main():
    lib = _lib.load(require_gpu=True)
    dev = torch.device("cuda")
    variants = [int(v) for v in sys.argv[1:]] or [0, 16, 1, 2, 3, 4, 7]
    for name, ci, co, H, W in (("conv1b", 64, 64, 540, 720), ("conv3b", 128, 128, 135, 180)):
        B = 32
        x = torch.randn(B, ci, H, W, device=dev)
        w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        u = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, dtype=torch.float32, device=dev)
        lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr())
        y = torch.empty(B, co, H // 2, W // 2, device=dev)
        for v in variants:
            os.environ["MFR_WINO_ABL"] = str(v)

            def run():
                rc = lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), None, B, ci, co, H, W, 1, 1, _lib.ptr(y), _lib.stream_ptr())
                assert rc == 0, rc
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            gf = 2 * 9 * ci * co * H * W * B / 1e9 / 2.25
            print(f"{name} ABL={v:2d}: {ms:7.3f} ms  {gf / ms:6.1f} TF (MFMA pipe)", flush=True)
        os.environ["MFR_WINO_ABL"] = "0"


if __name__ == "__main__":
    main()
