"""one `conv` block of the regression decoder under bf16 autocast on the GPU: asserts that the forward went through csrc/conv_gemm_bf16.hip
(regression/conv_bf16.py) and matches the library convolution.  python tools/check_decoder_conv_dispatch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mapfree_reloc_amd  # noqa: E402,F401
from mapfree_reloc_amd.regression import conv_bf16 as CB, encoder as E  # noqa: E402

calls = []
orig = CB.seg_gemm
CB.seg_gemm = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
m = E.conv(64, 64, 3, 1).cuda().train()
x = torch.randn(2, 64, 20, 12, device="cuda")
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = m(x)
    CB.ENABLED = False
    y2 = m(x)
torch.cuda.synchronize()
assert calls, "the own convolution kernel was not used under bf16 autocast"
assert float((y.float() - y2.float()).abs().max()) < 0.05, float((y.float() - y2.float()).abs().max())
print("DISPATCH OK", len(calls), float((y.float() - y2.float()).abs().max()))
