"""-m gpu parity: the implicit-GEMM convolution kernel (csrc/gemm_split.hip mfr_conv_igemm_f16x2: strided / 1x1 / 7x7 convolutions of the matcher
backbones on NCHW images, f16x2 arithmetic) vs a float64 convolution -- the layer shapes of LoFTR's ResNet-FPN (conv1 7x7 / 2, the stride-2 3x3 and
1x1 of layer2.0 / layer3.0, the FPN's 1x1 convolutions) and ragged ones (Cin not a multiple of 32, Cout not a multiple of 128, odd sizes, one pixel),
at the tolerance of the Winograd kernels' tests (2e-5 at unit-scale activations), plus the error class against the library's fp32 convolution."""
import pytest
import torch
import torch.nn.functional as F

from mapfree_reloc_amd.nets.conv import IgemmConv

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,ci,co,H,W,k,stride,pad,relu,bias", [
    (2, 1, 128, 64, 48, 7, 2, 3, 1, 1), (1, 1, 128, 720, 544, 7, 2, 3, 1, 1), (2, 128, 196, 90, 68, 3, 2, 1, 1, 1), (1, 196, 256, 45, 34, 3, 2, 1, 1, 1),
    (2, 128, 196, 90, 68, 1, 2, 0, 0, 1), (2, 256, 256, 23, 17, 1, 1, 0, 0, 1), (1, 196, 256, 45, 34, 1, 1, 0, 0, 0), (3, 5, 7, 9, 11, 3, 2, 1, 0, 1),
    (1, 33, 130, 13, 1, 3, 1, 1, 1, 1), (2, 64, 64, 1, 1, 1, 1, 0, 0, 1), (1, 40, 300, 17, 19, 5, 3, 2, 1, 0), (1, 1, 16, 10, 10, 3, 1, 1, 0, 1),
    (1, 128, 196, 360, 272, 3, 2, 1, 1, 1)])
def test_igemm_conv_vs_float64(B, ci, co, H, W, k, stride, pad, relu, bias):
    g = torch.Generator().manual_seed(B * 100 + ci + co + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, k, k, generator=g) / (k * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV) if bias else None
    y = IgemmConv(w, b, stride, pad)(x, relu=bool(relu))
    want = F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), stride=stride, padding=pad)
    if relu:
        want = want.relu()
    assert y.shape == want.shape and torch.isfinite(y).all()
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


def test_igemm_conv_error_class_is_fp32():
    """error against float64, normalised by sum |x||w| per output, at activation scales 1e-3 .. 1e3 and with output channels of uneven magnitude:
    within the class of the exact-fp32 matrix instruction (rms 2.8e-8, max 2.6e-7 of sum |x||w|: profiles/r05_f16x2_probe.jsonl) and, in rms, within 2x of
    the library's fp32 convolution on the same layer (MIOpen's direct kernel sums in shorter fp32 chains: measured rms 1.4e-8 / max 1.1e-7 against
    1.8e-8 / 1.8e-7 here)"""
    g = torch.Generator().manual_seed(3)
    for scale in (1.0, 1e-3, 1e3):
        x = (torch.randn(2, 128, 90, 68, generator=g) * scale).to(DEV)
        w = (torch.randn(196, 128, 3, 3, generator=g) / 34.0).to(DEV)
        w[::3] *= 1e-3
        want = F.conv2d(x.double().cpu(), w.double().cpu(), None, stride=2, padding=1)
        norm = F.conv2d(x.double().cpu().abs(), w.double().cpu().abs(), None, stride=2, padding=1) + 1e-300
        e2 = (IgemmConv(w, None, 2, 1)(x).double().cpu() - want) / norm
        e1 = (F.conv2d(x, w, None, stride=2, padding=1).double().cpu() - want) / norm
        rec = (scale, float(e2.abs().max()), float(e1.abs().max()), float(e2.pow(2).mean().sqrt()), float(e1.pow(2).mean().sqrt()))
        assert e2.abs().max() <= 3e-7 and e2.pow(2).mean().sqrt() <= 3e-8, rec
        assert e2.pow(2).mean().sqrt() <= 2.0 * e1.pow(2).mean().sqrt(), rec          # (the maxima are single outliers of 5e6 outputs: 1.8 - 2.3e-7 vs 1.1e-7)


@pytest.mark.parametrize("B,ci,co,Hl,Wl", [(2, 128, 196, 45, 34), (1, 196, 256, 23, 17), (2, 40, 70, 5, 7), (1, 32, 128, 1, 1)])
def test_igemm_conv_with_fpn_upsample_add(B, ci, co, Hl, Wl):
    """mfr_conv_igemm_f16x2_upadd: layerN_outconv(x) + F.interpolate(lo, scale_factor=2, bilinear, align_corners=True) in one launch (LoFTR's FPN merge,
    upstream ResNetFPN_8_2.forward) vs float64, and vs the two-launch path it replaces (convolution, then mfr_upsample2x_add)"""
    from mapfree_reloc_amd import _lib
    g = torch.Generator().manual_seed(B + ci + Hl)
    H, W = 2 * Hl, 2 * Wl
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    lo = torch.randn(B, co, Hl, Wl, generator=g).to(DEV)
    w = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).to(DEV)
    conv = IgemmConv(w, None, 1)
    y = conv(x, up_add=lo)
    want = F.conv2d(x.double().cpu(), w.double().cpu()) + F.interpolate(lo.double().cpu(), scale_factor=2.0, mode="bilinear", align_corners=True)
    assert (y.double().cpu() - want).abs().max().item() < 5e-5          # (the source coordinate is formed in f32, as in torch's kernel: tests/test_gpu_loftr_parity.py)
    two = conv(x)
    _lib.check(_lib.load().mfr_upsample2x_add(_lib.ptr(lo), _lib.ptr(two), B * co, Hl, Wl, _lib.stream_ptr()), "mfr_upsample2x_add")
    assert (y - two).abs().max().item() < 2e-6                           # same taps, same weights; only the compiler's contraction of the 7 operations may differ
