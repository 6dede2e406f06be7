"""-m gpu parity: fused Winograd F(2x2,3x3) convolution on the BF16 matrix cores at fp32 accuracy (csrc/winograd_bf16x3.hip: exact
3-way bf16 operand split, six partial products, fp32 accumulate) vs a float64 direct convolution -- the SAME shapes and the SAME
tolerance (|err| <= 2e-5 at unit-scale activations) as the exact-fp32 kernel's tests (tests/test_gpu_winograd_conv.py), plus an
error-class comparison against that kernel."""
import pytest
import torch
import torch.nn.functional as F

from mapfree_reloc_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pack(w):
    lib = _lib.load(require_gpu=True)
    co, ci = int(w.shape[0]), int(w.shape[1])
    nbytes = lib.mfr_wino_bf16x3_filter_bytes(ci, co)
    assert nbytes == -(-co // 64) * -(-ci // 16) * 96 * 1024
    u = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(lib.mfr_wino_bf16x3_filter_transform(_lib.ptr(w), ci, co, _lib.ptr(u), _lib.stream_ptr()), "filter")
    return u


VARIANTS = {"default": 0, "one_wave_per_simd": 32, "two_workgroups_per_cu": 2, "eight_wavefronts": 3}      # mfr_conv3x3_wino_bf16x3_variant


def _conv(x, w, b, act, pool, residual=None, variant=0):
    lib = _lib.load(require_gpu=True)
    B, ci, H, W = x.shape
    co = w.shape[0]
    u = _pack(w)
    y = torch.full((B, co, H // 2, W // 2) if pool else (B, co, H, W), float("nan"), dtype=torch.float32, device=x.device)
    _lib.check(lib.mfr_conv3x3_wino_bf16x3_variant(_lib.ptr(x), _lib.ptr(u), _lib.ptr(b) if b is not None else None,
                                                   _lib.ptr(residual) if residual is not None else None, B, ci, co, H, W, int(act), int(pool),
                                                   int(variant), _lib.ptr(y), _lib.stream_ptr()), "conv")
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
    (1, 256, 196, 45, 34, 2, 1, 1, 0), (1, 12, 40, 7, 9, 0, 0, 1, 1), (1, 128, 196, 136, 180, 1, 0, 1, 1)])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_bf16x3_conv_vs_float64(B, ci, co, H, W, act, pool, bias, res, variant):
    g = torch.Generator().manual_seed(B * 1000 + ci + H)
    x = torch.randn(B, ci, H, W, generator=g).to(DEV)
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
    b = torch.randn(co, generator=g).to(DEV) if bias else None
    r = torch.randn(B, co, H, W, generator=g).to(DEV) if res else None
    y = _conv(x, w, b, act, pool, r, VARIANTS[variant])
    want = _ref(x, w, b, act, pool, r)
    assert y.shape == want.shape
    assert torch.isfinite(y).all()                       # every output element written
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


def test_bf16x3_error_class_equals_exact_fp32_kernel():
    """max and rms error against float64 within 1.5x of the exact-fp32 matrix-core kernel's on the same layer, for unit-scale and
    for badly scaled activations (1e-3, 1e3) -- the split is exact, so scale must not matter"""
    g = torch.Generator().manual_seed(11)
    for scale in (1.0, 1e-3, 1e3):
        x = (torch.randn(2, 128, 90, 68, generator=g) * scale).to(DEV)
        w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).to(DEV)
        want = _ref(x, w, None, 0, 0)
        e3 = (_conv(x, w, None, 0, 0).double().cpu() - want) / scale
        e1 = (_exact(x, w, None, 0, 0).double().cpu() - want) / scale
        assert e3.abs().max() <= 1.5 * e1.abs().max() and e3.pow(2).mean().sqrt() <= 1.5 * e1.pow(2).mean().sqrt(), \
            (scale, float(e3.abs().max()), float(e1.abs().max()))


def test_bf16x3_kernel_generations_agree_bitwise():
    """the two-workgroups-per-CU kernel (round 4) runs the SAME products in the SAME order per output as the one-wavefront-per-SIMD
    kernel (round 3): outputs are the same bits, on a full-size layer of every kind the networks use"""
    g = torch.Generator().manual_seed(5)
    for (B, ci, co, H, W, act, pool, res) in ((2, 64, 64, 720, 540, 1, 1, 0), (2, 64, 128, 180, 135, 1, 0, 0), (1, 196, 196, 360, 272, 2, 0, 1),
                                              (1, 256, 196, 90, 68, 1, 0, 1)):
        x = torch.randn(B, ci, H, W, generator=g).to(DEV)
        w = (torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)).to(DEV)
        b = torch.randn(co, generator=g).to(DEV)
        r = torch.randn(B, co, H, W, generator=g).to(DEV) if res else None
        y1, y2 = _conv(x, w, b, act, pool, r, 32), _conv(x, w, b, act, pool, r, 2)
        assert torch.equal(y1, y2)


def test_bf16x3_linearity_and_shift():
    g = torch.Generator().manual_seed(7)
    x1 = torch.randn(1, 64, 135, 180, generator=g).to(DEV); x2 = torch.randn(1, 64, 135, 180, generator=g).to(DEV)
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).to(DEV)
    y1, y2, y12 = _conv(x1, w, None, 0, 0), _conv(x2, w, None, 0, 0), _conv(0.5 * x1 + x2, w, None, 0, 0)
    assert (y12 - (0.5 * y1 + y2)).abs().max().item() < 2e-5
    xs = torch.roll(x1, shifts=(3, 5), dims=(2, 3))
    ys = _conv(xs, w, None, 0, 0)
    assert (ys[:, :, 5:-2, 7:-2] - y1[:, :, 2:-5, 2:-7]).abs().max().item() < 2e-5


def test_bf16x3_rejects_unsupported():
    lib = _lib.load(require_gpu=True)
    x = torch.zeros(1, 4, 4, 4, device=DEV)
    assert lib.mfr_conv3x3_wino_bf16x3(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 1, 1, 0, 1, _lib.ptr(x), None) != 0      # pool needs H, W >= 2
    assert lib.mfr_conv3x3_wino_bf16x3(_lib.ptr(x), _lib.ptr(x), None, _lib.ptr(x), 1, 4, 32, 4, 4, 0, 1, _lib.ptr(x), None) != 0  # residual + pool
    assert lib.mfr_conv3x3_wino_bf16x3(_lib.ptr(x), _lib.ptr(x), None, None, 1, 4, 32, 4, 4, 3, 0, _lib.ptr(x), None) != 0      # unknown act
