"""-m gpu parity: the direct implicit-GEMM 3x3 convolution with an LDS-staged halo tile (csrc/conv_direct.hip, mfr_conv3x3_direct_f16x2; round 6) vs a
float64 direct convolution -- the SAME shapes and the SAME tolerance (|err| <= 2e-5 at unit-scale activations) as the split-Winograd kernel's tests
(tests/test_gpu_winograd_split.py), plus the error-class comparison against the exact-fp32 matrix-core kernel and the range guard."""
import pytest
import torch
import torch.nn.functional as F

from mapfree_reloc_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _conv(x, w, b, act, pool, residual=None):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    nbytes = lib.mfr_conv3x3_direct_f16x2_filter_bytes(ci, co)
    assert nbytes == -(-co // 64) * -(-ci // 16) * 54 * 1024 + -(-co // 64) * 256
    u = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(lib.mfr_conv3x3_direct_f16x2_filter_pack(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "pack")
    y = torch.full((B, co, H // 2, W // 2) if pool else (B, co, H, W), float("nan"), dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_conv3x3_direct_f16x2(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None,
                                            _lib.ptr(residual) if residual is not None else None, B, ci, co, H, W, int(act), int(pool),
                                            _lib.ptr(y), _lib.stream_ptr()), "conv")
    return y


def _exact(x, w):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    u = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "filter")
    y = torch.empty((B, co, H, W), dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(u), None, None, B, ci, co, H, W, 0, 0, _lib.ptr(y), _lib.stream_ptr()), "conv")
    return y


def _ref(x, w, b, act, pool, residual=None):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), padding=1)
    if residual is not None:
        y = y + residual.double().cpu()
    y = y.relu() if act == 1 else F.leaky_relu(y, 0.01) if act == 2 else y
    return F.max_pool2d(y, 2, 2) if pool else y


@pytest.mark.parametrize("B,ci,co,H,W,act,pool,bias,res", [
    (1, 16, 64, 8, 32, 0, 0, 0, 0), (2, 8, 32, 11, 38, 1, 0, 1, 0), (1, 64, 64, 17, 45, 1, 1, 1, 0), (3, 12, 96, 9, 33, 0, 1, 1, 0),
    (1, 4, 32, 2, 2, 1, 1, 1, 0), (1, 4, 32, 1, 1, 0, 0, 1, 0), (1, 16, 32, 12, 31, 1, 1, 1, 0), (2, 20, 64, 40, 130, 1, 0, 1, 0),
    (1, 128, 256, 67, 90, 1, 0, 1, 0), (2, 64, 128, 135, 180, 1, 1, 1, 0), (1, 64, 64, 540, 720, 1, 1, 1, 0),
    (2, 128, 256, 90, 67, 1, 0, 1, 0), (2, 64, 128, 180, 135, 1, 0, 1, 0), (1, 128, 128, 180, 135, 1, 1, 1, 0), (3, 8, 64, 7, 5, 0, 0, 1, 0),
    (1, 16, 64, 6, 33, 1, 1, 1, 0), (2, 8, 128, 5, 1, 1, 0, 1, 0), (1, 64, 64, 31, 35, 1, 1, 0, 0),
    # LoFTR backbone shapes: 196-channel stages, LeakyReLU, residual
    (1, 196, 196, 23, 34, 1, 0, 1, 1), (2, 196, 128, 20, 17, 2, 0, 1, 0), (1, 128, 128, 30, 44, 1, 0, 1, 1), (2, 8, 5, 9, 10, 2, 0, 1, 1),
    (1, 256, 196, 45, 34, 2, 1, 1, 0), (1, 12, 40, 7, 9, 0, 0, 1, 1), (1, 128, 196, 136, 180, 1, 0, 1, 1),
    # an ODD number of 64-channel groups above 64 channels: the 128-channel workgroup's second pair lies beyond the packed filter
    (1, 16, 160, 9, 33, 1, 0, 1, 0), (2, 32, 136, 8, 10, 1, 1, 1, 0), (1, 64, 192, 21, 70, 2, 0, 1, 1),
    # several tiles in both directions, odd sizes, one K step / many K steps
    (2, 16, 64, 70, 100, 1, 0, 1, 1), (1, 48, 128, 37, 71, 2, 1, 1, 0), (1, 256, 256, 68, 90, 1, 0, 1, 1)])
def test_direct_conv_vs_float64(B, ci, co, H, W, act, pool, bias, res):
    g = torch.Generator().manual_seed(B * 1000 + ci + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV) if bias else None
    r = torch.randn(B, co, H, W, generator=g).to(DEV) if res else None
    y = _conv(x, w, b, act, pool, r)
    want = _ref(x, w, b, act, pool, r)
    assert y.shape == want.shape
    assert torch.isfinite(y).all()                       # every output element written
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


def test_direct_error_class_is_fp32():
    """error against float64, normalised by sum |x||w|: the fp32 class -- max < 3e-7 (the exact-fp32 Winograd kernel measures 1.3e-7, PyTorch's fp32
    direct convolution on the CPU -- the arithmetic the reference's networks run in, matchers.py:50,105 -- 0.8e-7; this kernel 2.1e-7: its 72 x 3
    sequential accumulator roundings per output against Winograd's 8 x 3 and the CPU's blocked sums) and rms < 2.5e-8 (measured 1.8e-8, the f16x2
    probe's 1.7 - 2.2e-8: profiles/r05_f16x2_probe.jsonl), within 4x / 2.5x of the two (maxima over 1.5 M outputs) and 2x of the CPU's rms; for unit-scale and for badly scaled activations (1e-3, 1e3) and for
    filters of uneven magnitude across output channels"""
    g = torch.Generator().manual_seed(11)
    for scale in (1.0, 1e-3, 1e3):
        x = (torch.randn(2, 128, 90, 68, generator=g) * scale).to(DEV)
        w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).to(DEV)
        w[::3] *= 1e-3; w[1::5] *= 100.0
        want = _ref(x, w, None, 0, 0)
        norm = F.conv2d(x.double().cpu().abs(), w.double().cpu().abs(), None, padding=1) + 1e-300
        e3 = (_conv(x, w, None, 0, 0).double().cpu() - want) / norm
        e1 = (F.conv2d(x.cpu(), w.cpu(), None, padding=1).double() - want) / norm
        e0 = (_exact(x, w).double().cpu() - want) / norm
        rec = (scale, float(e3.abs().max()), float(e1.abs().max()), float(e0.abs().max()), float(e3.pow(2).mean().sqrt()), float(e1.pow(2).mean().sqrt()))
        assert e3.abs().max() < 3e-7 and e3.pow(2).mean().sqrt() < 2.5e-8, rec
        assert e3.abs().max() <= 4.0 * e1.abs().max() and e3.abs().max() <= 2.5 * e0.abs().max() and e3.pow(2).mean().sqrt() <= 2.0 * e1.pow(2).mean().sqrt(), rec


def test_direct_conv_of_relu_sparse_and_tiny_activations():
    g = torch.Generator().manual_seed(13)
    x = torch.randn(1, 64, 40, 66, generator=g).relu() * torch.exp(torch.randn(1, 64, 40, 66, generator=g) * 1.5)
    x = x.to(DEV)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(DEV)
    want = _ref(x, w, None, 0, 0)
    y = _conv(x, w, None, 0, 0).double().cpu()
    norm = F.conv2d(x.double().cpu().abs(), w.double().cpu().abs(), None, padding=1) + 1e-30
    e = (y - want) / norm
    assert float(e.abs().max()) < 1e-6 and float(e.pow(2).mean().sqrt()) < 1e-7      # (the bounds of the Winograd kernel's test of the same name)


def test_direct_linearity_and_shift():
    g = torch.Generator().manual_seed(7)
    x1 = torch.randn(1, 64, 135, 180, generator=g).to(DEV); x2 = torch.randn(1, 64, 135, 180, generator=g).to(DEV)
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).to(DEV)
    y1, y2, y12 = _conv(x1, w, None, 0, 0), _conv(x2, w, None, 0, 0), _conv(0.5 * x1 + x2, w, None, 0, 0)
    assert (y12 - (0.5 * y1 + y2)).abs().max().item() < 2e-5
    xs = torch.roll(x1, shifts=(3, 5), dims=(2, 3))
    ys = _conv(xs, w, None, 0, 0)
    assert (ys[:, :, 5:-2, 7:-2] - y1[:, :, 2:-5, 2:-7]).abs().max().item() < 2e-5


def test_direct_rejects_unsupported():
    lib = _lib.load(require_gpu=True)
    x = torch.zeros(1, 4, 4, 4, device=DEV)
    fn = lib.mfr_conv3x3_direct_f16x2
    assert fn(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 1, 1, 0, 1, _lib.ptr(x), None) != 0      # pool needs H, W >= 2
    assert fn(_lib.ptr(x), _lib.ptr(x), None, _lib.ptr(x), 1, 4, 32, 4, 4, 0, 1, _lib.ptr(x), None) != 0  # residual + pool
    assert fn(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 4, 4, 3, 0, _lib.ptr(x), None) != 0      # unknown act
    assert fn(None, _lib.ptr(x), None, None, 1, 4, 32, 4, 4, 0, 0, _lib.ptr(x), None) != 0
    assert lib.mfr_conv3x3_direct_f16x2_filter_bytes(0, 4) == 0


def test_direct_conv_range_guard():
    """one out-of-range (|x| > 65504) / non-finite input element raises the bound flag; in-range data does not (csrc/guard.h)"""
    lib = _lib.load(require_gpu=True)
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(128, 32, 3, 3, generator=g) / 17.0).to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for bad, where in ((None, None), (1e5, (0, 0, 0, 0)), (float("inf"), (1, 31, 36, 69)), (float("nan"), (0, 17, 16, 32)), (-7e4, (1, 5, 35, 0))):
        x = torch.randn(2, 32, 37, 70, generator=g).to(DEV)
        if bad is not None:
            x[where] = bad
        flag.zero_()
        _lib.check(lib.mfr_f16x2_guard_bind(_lib.ptr(flag)), "bind")
        try:
            _conv(x, w, None, 1, 0)
        finally:
            _lib.check(lib.mfr_f16x2_guard_bind(None), "unbind")
        assert int(flag.item()) == (0 if bad is None else 1), (bad, where)


@pytest.mark.parametrize("B,ci,co,H,W,act,bias", [
    (1, 16, 64, 8, 64, 0, 0), (2, 8, 32, 11, 38, 1, 1), (1, 64, 64, 17, 45, 1, 1), (3, 12, 96, 9, 33, 0, 1), (1, 4, 32, 2, 2, 1, 1), (1, 4, 32, 1, 1, 0, 1),
    (2, 20, 300, 40, 130, 1, 1), (1, 128, 196, 272, 360, 1, 1), (2, 196, 256, 136, 180, 1, 1), (1, 128, 256, 67, 91, 2, 1), (2, 48, 196, 135, 181, 1, 0)])
def test_strided_direct_conv_vs_float64(B, ci, co, H, W, act, bias):
    """mfr_conv3x3s2_direct_f16x2 (stride 2, pad 1; LoFTR's layer2.0 / layer3.0 conv1 shapes among them) against a float64 convolution, 2e-5 like the
    stride-1 kernel; odd and even sizes (the last patch column / row is present or not), several tiles, more than 256 output channels"""
    lib = _lib.load(require_gpu=True)
    g = torch.Generator().manual_seed(B * 1000 + ci + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV) if bias else None
    u = torch.empty(lib.mfr_conv3x3_direct_f16x2_filter_bytes(ci, co), dtype=torch.uint8, device=DEV)
    _lib.check(lib.mfr_conv3x3_direct_f16x2_filter_pack(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "pack")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.full((B, co, Ho, Wo), float("nan"), dtype=torch.float32, device=DEV)
    _lib.check(lib.mfr_conv3x3s2_direct_f16x2(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None, B, ci, co, H, W, act, _lib.ptr(y), _lib.stream_ptr()), "conv")
    want = F.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), stride=2, padding=1)
    want = want.relu() if act == 1 else F.leaky_relu(want, 0.01) if act == 2 else want
    assert y.shape == want.shape and torch.isfinite(y).all()
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,ci,co,H,W,act,ld", [(1, 16, 64, 8, 64, 0, 64), (2, 128, 256, 67, 90, 1, 256), (1, 196, 128, 41, 52, 0, 128), (2, 8, 36, 11, 38, 2, 48), (1, 64, 196, 20, 33, 1, 196)])
def test_direct_conv_rows_output_equals_nchw_output_bitwise(B, ci, co, H, W, act, ld):
    """mfr_conv3x3_direct_f16x2_rows: the token-major output is the NCHW output permuted, bit for bit (same accumulators, same epilogue arithmetic), incl. a row
    stride larger than Cout (columns beyond Cout untouched) and a 196-channel layer (3 x 2 blocking of the last group)"""
    lib = _lib.load(require_gpu=True)
    g = torch.Generator().manual_seed(ci + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV)
    want = _conv(x, w, b, act, 0)
    u = torch.empty(lib.mfr_conv3x3_direct_f16x2_filter_bytes(ci, co), dtype=torch.uint8, device=DEV)
    _lib.check(lib.mfr_conv3x3_direct_f16x2_filter_pack(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "pack")
    y = torch.full((B, H, W, ld), 7.0, dtype=torch.float32, device=DEV)
    _lib.check(lib.mfr_conv3x3_direct_f16x2_rows(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), B, ci, co, H, W, act, _lib.ptr(y), ld, _lib.stream_ptr()), "rows")
    assert torch.equal(y[..., :co].permute(0, 3, 1, 2), want)
    assert (y[..., co:] == 7.0).all()
    assert lib.mfr_conv3x3_direct_f16x2_rows(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b), B, ci, co, H, W, act, _lib.ptr(y), co - 4, None) != 0      # ldy < Cout
