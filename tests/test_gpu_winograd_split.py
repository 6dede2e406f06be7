"""-m gpu parity: fused Winograd F(2x2,3x3) convolution on the 16-bit matrix cores at fp32 accuracy by operand splitting
(csrc/winograd_split.hip; both arithmetics: 'f16x2', the default -- three partial products -- and 'bf16x3' -- six) vs a float64 direct
convolution -- the SAME shapes and the SAME tolerance (|err| <= 2e-5 at unit-scale activations) as the exact-fp32 kernel's tests
(tests/test_gpu_winograd_conv.py), plus an error-class comparison against that kernel at activation scales 1e-3 .. 1e3."""
import pytest
import torch
import torch.nn.functional as F

from mapfree_reloc_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SPLITS = ("f16x2", "bf16x3")


def _pack(w, split):
    lib = _lib.load(require_gpu=True)
    co, ci = int(w.shape[0]), int(w.shape[1])
    nbytes = getattr(lib, f"mfr_wino_{split}_filter_bytes")(ci, co)
    assert nbytes == -(-co // 64) * -(-ci // 16) * 96 * 1024 + (-(-co // 64) * 256 if split == "f16x2" else 0)
    u = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(getattr(lib, f"mfr_wino_{split}_filter_transform")(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "filter")
    return u


def _conv(x, w, b, act, pool, residual=None, split="f16x2"):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    u = _pack(w, split)
    y = torch.full((B, co, H // 2, W // 2) if pool else (B, co, H, W), float("nan"), dtype=torch.float32, device=x.device)
    _lib.check(getattr(lib, f"mfr_conv3x3_wino_{split}")(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None,
                                                         _lib.ptr(residual) if residual is not None else None, B, ci, co, H, W, int(act), int(pool),
                                                         _lib.ptr(y), _lib.stream_ptr()), "conv")
    return y


def _exact(x, w, b, act, pool, residual=None):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    u = torch.empty(lib.mfr_wino_filter_bytes(ci, co) // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_wino_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "filter")
    y = torch.empty((B, co, H // 2, W // 2) if pool else (B, co, H, W), dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_conv3x3_wino(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None,
                                    _lib.ptr(residual) if residual is not None else None, B, ci, co, H, W, int(act), int(pool), _lib.ptr(y),
                                    _lib.stream_ptr()), "conv")
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
    # more than 64 output channels with an ODD number of 64-channel groups (the 128-channel workgroup of the f16x2 kernel reads one group beyond the packed filter)
    (1, 16, 160, 9, 33, 1, 0, 1, 0), (2, 32, 136, 8, 10, 1, 1, 1, 0), (1, 64, 192, 21, 70, 2, 0, 1, 1)])
@pytest.mark.parametrize("split", SPLITS)
def test_split_conv_vs_float64(B, ci, co, H, W, act, pool, bias, res, split):
    g = torch.Generator().manual_seed(B * 1000 + ci + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV) if bias else None
    r = torch.randn(B, co, H, W, generator=g).to(DEV) if res else None
    y = _conv(x, w, b, act, pool, r, split)
    want = _ref(x, w, b, act, pool, r)
    assert y.shape == want.shape
    assert torch.isfinite(y).all()                       # every output element written
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("split", SPLITS)
def test_split_error_class_equals_exact_fp32_kernel(split):
    """max and rms error against float64 within 1.5x of the exact-fp32 matrix-core kernel's on the same layer, for unit-scale and for badly
    scaled activations (1e-3, 1e3) and for filters of uneven magnitude across output channels -- bf16x3 is exact whatever the scale; f16x2
    carries the low V term scaled by 2^11 and a per-channel filter scale for exactly this"""
    g = torch.Generator().manual_seed(11)
    for scale in (1.0, 1e-3, 1e3):
        x = (torch.randn(2, 128, 90, 68, generator=g) * scale).to(DEV)
        w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).to(DEV)
        w[::3] *= 1e-3; w[1::5] *= 100.0
        want = _ref(x, w, None, 0, 0)
        norm = F.conv2d(x.double().cpu().abs(), w.double().cpu().abs(), None, padding=1) + 1e-300
        e3 = (_conv(x, w, None, 0, 0, split=split).double().cpu() - want) / norm
        e1 = (_exact(x, w, None, 0, 0).double().cpu() - want) / norm
        assert e3.abs().max() <= 1.5 * e1.abs().max() and e3.pow(2).mean().sqrt() <= 1.5 * e1.pow(2).mean().sqrt(), \
            (split, scale, float(e3.abs().max()), float(e1.abs().max()), float(e3.pow(2).mean().sqrt()), float(e1.pow(2).mean().sqrt()))


def test_f16x2_conv_of_relu_sparse_and_tiny_activations():
    """what the layers actually see: half the activations exactly zero (ReLU), the rest spanning five decades (1e-3 .. 4e3; the f16 term needs |activation| < 16376) -- subnormal f16 terms must
    survive the matrix instruction"""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(1, 64, 40, 66, generator=g).relu() * torch.exp(torch.randn(1, 64, 40, 66, generator=g) * 1.5)
    assert float(x.max()) < 16000
    x = x.to(DEV)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(DEV)
    want = _ref(x, w, None, 0, 0)
    norm = F.conv2d(x.double().cpu().abs(), w.double().cpu().abs(), None, padding=1) + 1e-30
    e = (_conv(x, w, None, 0, 0, split="f16x2").double().cpu() - want) / norm
    assert float(e.abs().max()) < 1e-6 and float(e.pow(2).mean().sqrt()) < 1e-7


@pytest.mark.parametrize("split", SPLITS)
def test_split_linearity_and_shift(split):
    g = torch.Generator().manual_seed(7)
    x1 = torch.randn(1, 64, 135, 180, generator=g).to(DEV); x2 = torch.randn(1, 64, 135, 180, generator=g).to(DEV)
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).to(DEV)
    y1, y2, y12 = _conv(x1, w, None, 0, 0, split=split), _conv(x2, w, None, 0, 0, split=split), _conv(0.5 * x1 + x2, w, None, 0, 0, split=split)
    assert (y12 - (0.5 * y1 + y2)).abs().max().item() < 2e-5
    xs = torch.roll(x1, shifts=(3, 5), dims=(2, 3))
    ys = _conv(xs, w, None, 0, 0, split=split)
    assert (ys[:, :, 5:-2, 7:-2] - y1[:, :, 2:-5, 2:-7]).abs().max().item() < 2e-5


@pytest.mark.parametrize("split", SPLITS)
def test_split_rejects_unsupported(split):
    lib = _lib.load(require_gpu=True)
    x = torch.zeros(1, 4, 4, 4, device=DEV)
    fn = getattr(lib, f"mfr_conv3x3_wino_{split}")
    assert fn(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 1, 1, 0, 1, _lib.ptr(x), None) != 0      # pool needs H, W >= 2
    assert fn(_lib.ptr(x), _lib.ptr(x), None, _lib.ptr(x), 1, 4, 32, 4, 4, 0, 1, _lib.ptr(x), None) != 0  # residual + pool
    assert fn(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 4, 4, 3, 0, _lib.ptr(x), None) != 0      # unknown act


@pytest.mark.parametrize("B,H,W", [(2, 720, 540), (1, 64, 96), (3, 37, 45), (1, 8, 32), (2, 2, 2), (1, 131, 33)])
def test_fused_conv1a_conv1b_equals_the_two_launches_bitwise(B, H, W):
    """mfr_sp_conv1ab_f16x2 (conv1a + ReLU computed into conv1b's LDS patches) == mfr_conv3x3_c1_relu followed by mfr_conv3x3_wino_f16x2(act 1,
    pool 1), bit for bit: interior and border workgroups, odd sizes, images smaller than one workgroup block"""
    lib = _lib.load(require_gpu=True)
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.rand(B, 1, H, W, generator=g).to(DEV)
    w1 = (torch.randn(64, 1, 3, 3, generator=g) / 3.0).to(DEV); b1 = (torch.randn(64, generator=g) * 0.3).to(DEV)
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(DEV); b2 = torch.randn(64, generator=g).to(DEV)
    u = _pack(w2, "f16x2")
    fused = torch.full((B, 64, H // 2, W // 2), float("nan"), dtype=torch.float32, device=DEV)
    _lib.check(lib.mfr_sp_conv1ab_f16x2(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(u), _lib.ptr(b2), B, H, W, _lib.ptr(fused), _lib.stream_ptr()), "fused")
    if W % 4 == 0:
        mid = torch.empty(B, 64, H, W, dtype=torch.float32, device=DEV)
        _lib.check(lib.mfr_conv3x3_c1_relu(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), B, H, W, 64, _lib.ptr(mid), _lib.stream_ptr()), "conv1a")
    else:                                                   # the stand-alone first-layer kernel wants W % 4 == 0: the same chain of fmas in torch
        xp = F.pad(x[:, 0], (1, 1, 1, 1))
        mid = torch.zeros(B, 64, H, W, dtype=torch.float32, device=DEV)
        for dy in range(3):
            for dx in range(3):
                mid = torch.addcmul(mid, w1[:, 0, dy, dx].view(1, 64, 1, 1), xp[:, None, dy:dy + H, dx:dx + W])      # fma per tap, dy-major like the kernel
        mid = (mid + b1.view(1, 64, 1, 1)).relu()
    two = _conv(mid, w2, b2, 1, 1, None, "f16x2")
    assert torch.isfinite(fused).all()
    if W % 4 == 0:
        assert torch.equal(fused, two)
    else:                                                   # (torch's addcmul is not guaranteed to contract into one fma: tolerance instead of bits)
        assert (fused - two).abs().max().item() < 1e-4
    want = _ref(_ref(x, w1, b1, 1, 0).float().to(DEV), w2, b2, 1, 1)
    assert (fused.double().cpu() - want).abs().max().item() < 1e-4
