"""Host-side layouts of regression/conv_bf16.py (haloed NHWC / channel-major images, tap shifts, segment tables, K splits) checked on
the CPU: the kernel launch is replaced by tests/seg_gemm_emul.py, a numpy statement of the kernel's address arithmetic, and forward,
d input, d weight, d bias are compared with torch's own convolution on bf16-rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mapfree_reloc_amd  # noqa: F401
from mapfree_reloc_amd.regression import conv_bf16 as CB
from tests import seg_gemm_emul


@pytest.mark.parametrize("B,C,N,H,W,splits", [(2, 32, 32, 6, 5, None), (1, 64, 32, 5, 9, 3), (3, 32, 64, 4, 4, 2)])
def test_conv3x3_bf16_layouts_match_torch(monkeypatch, B, C, N, H, W, splits):
    monkeypatch.setattr(CB, "seg_gemm", seg_gemm_emul.seg_gemm)
    monkeypatch.setattr(CB, "BACKWARD", "hip")
    monkeypatch.setattr(CB, "pack_nhwc_halo", seg_gemm_emul.pack_nhwc_halo)
    monkeypatch.setattr(CB, "pack_cm_halo", seg_gemm_emul.pack_cm_halo)
    monkeypatch.setattr(CB, "unpack_nchw", seg_gemm_emul.unpack_nchw)
    if splits:
        orig = CB._wgrad
        monkeypatch.setattr(CB, "_wgrad", lambda x, gy: orig(x, gy, splits=splits))
    g = torch.Generator().manual_seed(B * 100 + C + N + H)
    r = lambda *s: torch.randn(*s, generator=g).bfloat16().float()          # bf16-representable values: products are exact in fp32
    x, w, b, gy = r(B, C, H, W), r(N, C, 3, 3) * 0.1, r(N), r(B, N, H, W)
    x1, w1, b1 = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y1 = CB.conv3x3_bf16(x1, w1, b1)
    assert y1.shape == (B, N, H, W) and y1.dtype == torch.bfloat16
    y1.backward(gy.bfloat16())
    x2, w2, b2 = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y2 = F.conv2d(x2, w2, b2, padding=1)
    y2.backward(gy)
    tol = lambda ref: 1e-2 * float(ref.abs().max())                          # one bf16 rounding of the result
    assert float((y1.float() - y2).abs().max()) <= tol(y2)
    assert float((x1.grad - x2.grad).abs().max()) <= tol(x2.grad)            # d input leaves the kernel in bf16
    assert float((w1.grad - w2.grad).abs().max()) <= 1e-4 * float(w2.grad.abs().max()) + 1e-5      # d weight is fp32 end to end
    assert float((b1.grad - b2.grad).abs().max()) <= 1e-4 * float(b2.grad.abs().max()) + 1e-5
    assert x1.grad.dtype == torch.float32 and w1.grad.dtype == torch.float32
