"""-m gpu parity: fused Winograd F(2x2,3x3) convolution on the fp32 matrix cores (csrc/winograd_conv.hip)
vs a float64 direct convolution (torch CPU), incl. the fused bias / ReLU / 2x2 max-pool epilogue, ragged and
odd shapes (zero padding via out-of-range buffer reads, odd widths take the non-paired load path), and the
SuperPoint layer shapes.  Tolerance: f32 Winograd arithmetic, |err| <= 2e-5 for unit-scale activations
(the library's own f32 convolution is within ~8e-6 of the same reference)."""
import pytest
import torch
import torch.nn.functional as F

from mapfree_reloc_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _wino(x, w, b, relu, pool, residual=None, variant=0):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    nbytes = lib.mfr_wino_filter_bytes(ci, co)
    assert nbytes == 16 * ci * (-(-co // 32) * 32) * 4
    u = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "filter")
    y = torch.full((B, co, H // 2, W // 2) if pool else (B, co, H, W), float("nan"), dtype=torch.float32, device=x.device)
    args = (_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None, _lib.ptr(residual) if residual is not None else None,
            B, ci, co, H, W, int(relu), int(pool))
    if variant == 0:
        _lib.check(lib.mfr_conv3x3_wino(*args, _lib.ptr(y), _lib.stream_ptr()), "conv")
    else:
        _lib.check(lib.mfr_conv3x3_wino_variant(*args, variant, _lib.ptr(y), _lib.stream_ptr()), "conv")
    return y


def _ref(x, w, b, act, pool, residual=None):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)
    if residual is not None:
        y = y + residual.double().cpu()
    y = y.relu() if act == 1 else F.leaky_relu(y, 0.01) if act == 2 else y
    return F.max_pool2d(y, 2, 2) if pool else y


@pytest.mark.parametrize("B,ci,co,H,W,relu,pool,bias", [
    (1, 4, 32, 8, 32, 0, 0, 0), (2, 8, 32, 11, 38, 1, 0, 1), (1, 64, 64, 17, 45, 1, 1, 1), (3, 12, 96, 9, 33, 0, 1, 1),
    (1, 4, 32, 2, 2, 1, 1, 1), (1, 4, 32, 1, 1, 0, 0, 1), (1, 16, 32, 12, 31, 1, 1, 1), (2, 20, 64, 40, 130, 1, 0, 1),
    (1, 128, 256, 67, 90, 1, 0, 1), (2, 64, 128, 135, 180, 1, 1, 1), (1, 64, 64, 540, 720, 1, 1, 1),
    # odd widths (linear tiling across row breaks): the SuperPoint layers at 1/4 and 1/8 resolution of a 720x540 image, + small ones
    (2, 128, 256, 90, 67, 1, 0, 1), (2, 64, 128, 180, 135, 1, 0, 1), (1, 128, 128, 180, 135, 1, 1, 1), (3, 8, 64, 7, 5, 0, 0, 1),
    (1, 16, 64, 6, 33, 1, 1, 1), (2, 8, 128, 5, 1, 1, 0, 1), (1, 64, 64, 31, 35, 1, 1, 0)])
def test_wino_conv_vs_float64(B, ci, co, H, W, relu, pool, bias):
    g = torch.Generator().manual_seed(B * 1000 + ci + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV) if bias else None
    y = _wino(x, w, b, relu, pool)
    r = _ref(x, w, b, relu, pool)
    assert y.shape == r.shape
    assert torch.isfinite(y).all()                       # every output element written
    assert (y.double().cpu() - r).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,ci,co,H,W,act,pool,res", [
    (2, 64, 64, 30, 52, 1, 1, 0), (1, 64, 128, 45, 67, 1, 0, 0), (1, 196, 196, 23, 34, 1, 0, 1), (1, 12, 40, 7, 9, 2, 0, 1),
    (2, 4, 32, 5, 3, 0, 0, 0), (1, 132, 256, 19, 33, 1, 1, 0), (2, 64, 64, 23, 67, 1, 0, 1), (1, 64, 128, 40, 135, 2, 0, 1)])
def test_wino_kernel_variants_agree(B, ci, co, H, W, act, pool, res):
    """the three kernels behind mfr_conv3x3_wino (classic, software-pipelined, shared-transform) on the same operands: the
    pipelined one is bit-identical to the classic one (same arithmetic, different schedule); the shared-transform one folds
    the bias into an accumulator and carries one Winograd row negated -- f32-roundoff differences only.  Odd chunk counts
    (Cin % 8 == 4), odd widths and padded cout blocks included."""
    g = torch.Generator().manual_seed(ci + 13 * co + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    r = torch.randn(B, co, H, W, generator=g).to(DEV) if res else None
    y1, y2, y4 = (_wino(x, w, b, act, pool, r, variant=v) for v in (1, 2, 4))
    want = _ref(x, w, b, act, pool, r)
    assert torch.equal(y1, y2)
    for y in (y1, y4):
        assert torch.isfinite(y).all() and (y.double().cpu() - want).abs().max().item() < 2e-5


def test_wino_linearity_and_shift():
    """size-independent properties at a SuperPoint layer shape: conv(a x1 + x2) = a conv(x1) + conv(x2) (no bias/ReLU)
    and a 2-pixel shift of the input shifts the output by 2 pixels away from the borders (tile-grid independence)"""
    g = torch.Generator().manual_seed(7)
    x1 = torch.randn(1, 64, 135, 180, generator=g).to(DEV); x2 = torch.randn(1, 64, 135, 180, generator=g).to(DEV)
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).to(DEV)
    y1, y2, y12 = _wino(x1, w, None, 0, 0), _wino(x2, w, None, 0, 0), _wino(0.5 * x1 + x2, w, None, 0, 0)
    assert (y12 - (0.5 * y1 + y2)).abs().max().item() < 2e-5
    xs = torch.roll(x1, shifts=(3, 5), dims=(2, 3))      # odd shifts: different Winograd tile phase
    ys = _wino(xs, w, None, 0, 0)
    assert (ys[:, :, 5:-2, 7:-2] - y1[:, :, 2:-5, 2:-7]).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,ci,co,H,W,act,pool,res", [
    (1, 196, 196, 23, 34, 1, 0, 0), (2, 196, 128, 20, 17, 2, 0, 0), (1, 128, 128, 30, 44, 1, 0, 1), (2, 8, 5, 9, 10, 2, 0, 1),
    (1, 256, 196, 45, 34, 2, 1, 0), (1, 12, 40, 7, 9, 0, 0, 1)])
def test_wino_padded_cout_residual_leaky(B, ci, co, H, W, act, pool, res):
    """the LoFTR ResNet-FPN uses 196-channel stages (Cout padded to 224 inside the packed filter), LeakyReLU(0.01)
    in the FPN heads and `relu(x + conv(y))` at the end of every BasicBlock"""
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    r = torch.randn(B, co, H, W, generator=g).to(DEV) if res else None
    y = _wino(x, w, b, act, pool, r)
    want = _ref(x, w, b, act, pool, r)
    assert y.shape == want.shape and torch.isfinite(y).all()
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


def test_wino_rejects_unsupported():
    lib = _lib.load(require_gpu=True)
    assert lib.mfr_wino_filter_bytes(3, 32) == 0 and lib.mfr_wino_filter_bytes(4, 48) == 16 * 4 * 64 * 4
    x = torch.zeros(1, 4, 4, 4, device=DEV)
    assert lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(x), None, None, 1, 3, 32, 4, 4, 0, 0, _lib.ptr(x), None) != 0
    assert lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 1, 1, 0, 1, _lib.ptr(x), None) != 0  # pool needs H,W >= 2
    assert lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(x), None, _lib.ptr(x), 1, 4, 32, 4, 4, 0, 1, _lib.ptr(x), None) != 0  # residual + pool
    assert lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 4, 4, 3, 0, _lib.ptr(x), None) != 0  # unknown act
