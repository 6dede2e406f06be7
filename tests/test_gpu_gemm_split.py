"""-m gpu: linear layers on the 16-bit matrix cores at fp32 accuracy (csrc/gemm_split.hip; both arithmetics: 'f16x2', the default, and
'bf16x3') vs a float64 product: the transformer shapes, strided in-place operands, ReLU / accumulate epilogues, ragged M and N, and the
error class against the library's fp32 GEMM at activation scales 1e-3 .. 1e3."""
import pytest
import torch

from mapfree_reloc_amd.nets.linear import SplitLinear

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SPLITS = ("f16x2", "bf16x3")


@pytest.mark.parametrize("M,K,N,relu,acc,bias", [
    (4096, 256, 768, 0, 0, 1), (4096, 512, 512, 1, 0, 1), (4096, 512, 256, 0, 1, 0), (1000, 256, 256, 0, 0, 1), (77, 32, 40, 1, 0, 1),
    (129, 64, 130, 0, 1, 1), (6120, 256, 512, 1, 0, 0), (1, 32, 1, 0, 0, 1), (300, 128, 128, 0, 0, 1)])
@pytest.mark.parametrize("split", SPLITS)
def test_split_linear_vs_float64(M, K, N, relu, acc, bias, split):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    y0 = torch.randn(M, N, generator=g).to(DEV)
    lin = SplitLinear(w, b, split=split)
    y = lin(x, out=y0.clone() if acc else None, relu=bool(relu), accumulate=bool(acc))
    want = x.double() @ w.double().t()
    if bias:
        want = want + b.double()
    if relu:
        want = want.relu()
    if acc:
        want = want + y0.double()
    assert y.shape == (M, N) and torch.isfinite(y).all()
    assert (y.double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,K,N,relu,acc", [(8192 + 77, 256, 768, 0, 0), (4096, 512, 512, 1, 0), (5000, 512, 256, 0, 1), (300, 128, 200, 1, 1), (129, 64, 130, 0, 1),
                                            (70000, 160, 256, 0, 0), (1000, 64, 100, 1, 0), (33000, 192, 384, 0, 1)])
@pytest.mark.parametrize("split", SPLITS)
def test_kernel_generations_agree_bitwise(M, K, N, relu, acc, split):
    """one tile per workgroup (round 3, flag 4), persistent 128x128 workgroups with register-staged W (flag 8) and the default (W by LDS-DMA, X
    two steps ahead; K % 64 != 0 runs flag 8): the same sums in the same order for every output element -- only where and when a tile is computed differs"""
    g = torch.Generator().manual_seed(M ^ N)
    x = torch.randn(M, K + 32, generator=g).to(DEV)[:, :K]                  # row stride > K
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    y0 = torch.randn(M, N + 8, generator=g).to(DEV)
    lin = SplitLinear(w, b, split=split)
    outs = []
    for fl in (4, 8, 0):
        y = y0.clone()
        yv = y[:, :N]
        lin(x, out=yv, relu=bool(relu), accumulate=bool(acc), kernel_flag=fl)
        torch.cuda.synchronize()
        assert torch.equal(y[:, N:], y0[:, N:])                             # nothing written beyond the N columns
        outs.append(yv.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = x.double() @ w.double().t() + b.double()
    if relu:
        want = want.relu()
    if acc:
        want = want + y0[:, :N].double()
    assert (outs[-1].double() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("split", SPLITS)
def test_split_linear_strided_in_place(split):
    """the SuperGlue layer's operands: x~ = left half of the [x~ | a] buffer (row stride 512), the MLP reads all 512 columns and
    its second layer accumulates into the left half in place"""
    g = torch.Generator().manual_seed(3)
    M = 2048
    xa = torch.randn(M, 512, generator=g).to(DEV)
    xv = xa[:, :256]
    wq = (torch.randn(768, 256, generator=g) / 16).to(DEV); w1 = (torch.randn(512, 512, generator=g) / 22).to(DEV)
    w2 = (torch.randn(256, 512, generator=g) / 22).to(DEV); bq = torch.randn(768, generator=g).to(DEV); b1 = torch.randn(512, generator=g).to(DEV)
    ref_q = xv.double() @ wq.double().t() + bq.double()
    ref_h = (xa.double() @ w1.double().t() + b1.double()).relu()
    ref_x = xv.double() + ref_h @ w2.double().t()
    q = SplitLinear(wq, bq, split=split)(xv)
    h = SplitLinear(w1, b1, split=split)(xa, relu=True)
    SplitLinear(w2, split=split)(h, out=xv, accumulate=True)
    assert (q.double() - ref_q).abs().max() < 2e-5 and (h.double() - ref_h).abs().max() < 2e-5
    assert (xv.double() - ref_x).abs().max() < 3e-5
    assert torch.equal(xa[:, 256:], xa[:, 256:])          # right half untouched (no NaN introduced)


@pytest.mark.parametrize("split", SPLITS)
def test_split_linear_error_class_vs_library_fp32(split):
    """the split product is in the error class of the library's fp32 GEMM whatever the activation scale (f16x2 carries its low term scaled
    by 2^11 for exactly this: unscaled it would be good to an ABSOLUTE 2^-25 only and fail at 1e-3), and for weights of uneven magnitude
    across output features (the per-feature scale of the f16x2 packing)"""
    g = torch.Generator().manual_seed(5)
    for scale in (1.0, 1e-3, 1e3):
        x = (torch.randn(8192, 512, generator=g) * scale).to(DEV)
        w = (torch.randn(512, 512, generator=g) / 22).to(DEV)
        w[::3] *= 1e-4; w[1::7] *= 300.0
        want = x.double() @ w.double().t()
        norm = x.double().abs() @ w.double().abs().t()
        e3 = (SplitLinear(w, split=split)(x).double() - want) / norm
        e1 = ((x @ w.t()).double() - want) / norm
        assert e3.abs().max() <= 1.5 * e1.abs().max() and e3.pow(2).mean().sqrt() <= 1.5 * e1.pow(2).mean().sqrt(), \
            (split, scale, float(e3.abs().max()), float(e1.abs().max()), float(e3.pow(2).mean().sqrt()), float(e1.pow(2).mean().sqrt()))


def test_f16x2_small_and_zero_operands():
    """subnormal f16 terms survive the matrix instruction: activations far below the f16 normal range, a zero weight row, zero activations"""
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(512, 256, generator=g) * 1e-6).to(DEV)
    x[:7] = 0.0
    w = (torch.randn(256, 256, generator=g) / 16).to(DEV)
    w[5] = 0.0
    y = SplitLinear(w, split="f16x2")(x)
    want = x.double() @ w.double().t()
    assert torch.isfinite(y).all() and (y[:7] == 0).all() and (y[:, 5] == 0).all()
    assert (y.double() - want).abs().max().item() < 1e-10                    # |x| ~ 1e-6: absolute floor 2^-36 per element, 256 terms of |w| ~ 0.06


def test_split_linear_rejects():
    with pytest.raises(ValueError):
        SplitLinear(torch.zeros(8, 48, device=DEV))


@pytest.mark.parametrize("nb,M,N,K,ldx,ldw,mul", [(32, 1024, 1024, 256, 256, 256, 1.0 / 16.0), (3, 1200, 1100, 256, 512, 512, 1.0 / 256.0), (2, 77, 130, 64, 64, 96, 1.0),
                                                  (1, 4800, 4800, 256, 512, 512, 1.0 / 256.0), (5, 1, 1, 32, 32, 32, 2.0)])
def test_batched_products_equal_the_per_problem_launches_and_float64(nb, M, N, K, ldx, ldw, mul):
    """the matchers' score / similarity matrices (SuperGlue S = mdesc0 mdesc1^T / 16 on interleaved pairs, LoFTR feat_c0 feat_c1^T / C on the row-strided
    halves of the [x | message] buffer): one batched launch == SplitLinear on every problem with the same out_mul applied afterwards (a power of two:
    bit for bit), and the fp32-class error against a float64 product"""
    from mapfree_reloc_amd.nets.linear import SplitBatchedNT
    g = torch.Generator().manual_seed(nb * 131 + M + N)
    xa = (torch.randn(2 * nb, M, ldx, generator=g) * 3.0).to(DEV)
    wa = (torch.randn(2 * nb, N, ldw, generator=g) * torch.rand(2 * nb, N, 1, generator=g) * 4.0).to(DEV)
    x, w = xa[0::2, :, :K], wa[1::2, :, :K]                                  # batch stride 2 x, row stride > K
    bg = SplitBatchedNT()
    y = bg(x, w, out_mul=mul)
    assert y.shape == (nb, M, N) and torch.isfinite(y).all()
    for b in range(nb):
        one = SplitLinear(w[b].contiguous(), None, split="f16x2")(x[b])
        assert torch.equal(one * mul, y[b])
        want = (x[b].double() @ w[b].double().t()) * mul
        err = (y[b].double() - want).abs()
        ref = (x[b] @ w[b].t() * mul).double()
        assert err.max().item() <= 4e-6 * want.abs().max().item() + 1e-12
        assert err.pow(2).mean().sqrt().item() <= 2.0 * (ref - want).pow(2).mean().sqrt().item() + 1e-12
    # the strided output form (written inside a wider buffer) and a second call on the same object (scratch reuse)
    big = torch.full((nb, M, N + 8), 7.0, device=DEV)
    bg(x, w, out_mul=mul, out=big[:, :, :N])
    assert torch.equal(big[:, :, :N], y) and (big[:, :, N:] == 7.0).all()


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("M,K", [(1, 128), (127, 128), (128, 256), (1031, 256), (50000, 128)])
def test_linear_plus_layernorm_in_one_launch(split, M, K):
    """mfr_gemm_*_ln (round 6): LayerNorm_128(x W^T + b) [+ residual, in place] in the GEMM epilogue -- the `norm1(merge(.))` / `x + norm2(mlp(.))` pairs of
    upstream LoFTREncoderLayer at d_model 128 -- against float64 and against the two launches it replaces (mfr_gemm_* then mfr_layernorm: same two-pass
    statistics, another summation order).  Strided operands as nets/loftr.py uses them (the [x | message] buffer); f32 tolerance 2e-6 (1 + |y|) as in
    test_fused_layernorm_vs_torch, on normalised outputs of unit scale."""
    import torch.nn.functional as F
    from mapfree_reloc_amd.nets.linear import SplitLinear
    from mapfree_reloc_amd.nets.loftr import LoFTRHIP
    g = torch.Generator().manual_seed(M + K)
    w = (torch.randn(128, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    gam, bet = (1 + 0.3 * torch.randn(128, generator=g)).to(DEV), torch.randn(128, generator=g).to(DEV)
    lin = SplitLinear(w, b, split=split)
    assert lin.ln_fusable()
    xin = torch.randn(M, K + 64, generator=g).to(DEV)[:, :K]                       # row stride K + 64
    xm = torch.randn(M, 256, generator=g).to(DEV)
    keep = xm.clone()
    want = F.layer_norm(xin.double().cpu() @ w.double().cpu().T + b.double().cpu(), (128,), gam.double().cpu(), bet.double().cpu(), 1e-5)
    lin(xin, out=xm[:, 128:], ln=(gam, bet))                                        # norm1 -> right half of [x | message]
    got = xm[:, 128:].double().cpu()
    assert torch.equal(xm[:, :128], keep[:, :128])
    assert ((got - want).abs() / (1 + want.abs())).max().item() < 4e-6
    two = torch.empty(M, 128, device=DEV)
    LoFTRHIP.layernorm(lin(xin), (gam, bet), two)
    assert ((got - two.double().cpu()).abs() / (1 + want.abs())).max().item() < 4e-6
    lin(xin, out=xm[:, :128], ln=(gam, bet), accumulate=True)                       # x += norm2(...) in place
    want2 = keep[:, :128].double().cpu() + want
    assert ((xm[:, :128].double().cpu() - want2).abs() / (1 + want2.abs())).max().item() < 4e-6
    assert torch.equal(xm[:, 128:].double().cpu(), got)                             # the right half is not touched by the in-place update


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("nwin,with_bias", [(1, True), (7, False), (500, True), (2049, True)])
def test_windowed_product_equals_gather_then_gemm_then_add(split, nwin, with_bias):
    """mfr_gemm_*_windows (round 6, upstream FinePreprocess: unfold -> gather at the matches -> merge_feat): the 5x5 windows' tokens read from the NHWC
    fine map inside the product + the per-window bias in the epilogue == mfr_loftr_gather_windows -> mfr_gemm_* -> broadcast add, bit for bit (same
    operands, same K order, the same two roundings); windows at the map's border exercise the zero padding, 2049 windows a partial last row block"""
    from mapfree_reloc_amd import _lib
    from mapfree_reloc_amd.nets.linear import SplitLinear
    lib = _lib.load()
    g = torch.Generator().manual_seed(nwin)
    Bimg, Hf, Wf, C, wc, hc, stride, W = 4, 60, 44, 128, 11, 15, 4, 5
    feat = torch.randn(Bimg, Hf, Wf, C, generator=g).to(DEV)
    img = torch.randint(0, Bimg, (nwin,), generator=g).int().to(DEV)
    cell = torch.randint(0, wc * hc, (nwin,), generator=g).int()
    cell[0] = 0; cell[-1] = wc * hc - 1                                            # corners: two rows / columns of padding
    cell = cell.to(DEV)
    lin = SplitLinear((torch.randn(128, C, generator=g) / C ** 0.5).to(DEV), split=split)
    cw = torch.randn(nwin, 128, generator=g).to(DEV) if with_bias else None
    xf = torch.full((nwin * W * W, 256), 7.0, device=DEV)
    lin.windows(feat, img, cell, wc, stride, W, out=xf[:, :128], window_bias=cw)
    win = torch.empty(nwin, W * W, C, device=DEV)
    _lib.check(lib.mfr_loftr_gather_windows(_lib.ptr(feat), Bimg, Hf, Wf, C, _lib.ptr(img), _lib.ptr(cell), nwin, wc, stride, W, _lib.ptr(win), _lib.stream_ptr()), "gather")
    want = lin(win.view(nwin * W * W, C)).view(nwin, W * W, 128)
    if cw is not None:
        want = want + cw[:, None, :]
    assert torch.equal(xf[:, :128], want.reshape(nwin * W * W, 128))
    assert (xf[:, 128:] == 7.0).all()


@pytest.mark.parametrize("split", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("M", [1, 127, 128, 1031, 40000])
def test_fused_mlp_layernorm_residual(split, M):
    """mfr_mlp_ln_* (round 6): x += LayerNorm_128(relu([x | msg] W1^T) W2^T) in ONE launch (upstream LoFTREncoderLayer: mlp -> norm2 -> residual, d_model
    128) against float64 and against the two launches it replaces (mfr_gemm_* with ReLU, then mfr_gemm_*_ln): in place on the left half of the [x | message]
    buffer as nets/loftr.py runs it; partial last row tile, single row, many tiles per workgroup (40000 rows = 313 tiles on a 512-workgroup grid)."""
    import torch.nn.functional as F
    from mapfree_reloc_amd.nets.linear import FusedMlpLn, SplitLinear
    g = torch.Generator().manual_seed(M)
    w1 = (torch.randn(256, 256, generator=g) / 16.0).to(DEV)
    w2 = (torch.randn(128, 256, generator=g) / 16.0).to(DEV)
    gam, bet = (1 + 0.3 * torch.randn(128, generator=g)).to(DEV), torch.randn(128, generator=g).to(DEV)
    l1, l2 = SplitLinear(w1, split=split), SplitLinear(w2, split=split)
    mlp = FusedMlpLn(l1, l2)
    xm = torch.randn(M, 256, generator=g).to(DEV)
    keep = xm.clone()
    hid64 = (keep.double().cpu() @ w1.double().cpu().T).relu()
    want = keep[:, :128].double().cpu() + F.layer_norm(hid64 @ w2.double().cpu().T, (128,), gam.double().cpu(), bet.double().cpu(), 1e-5)
    two = keep.clone()
    l2(l1(two, relu=True), out=two[:, :128], ln=(gam, bet), accumulate=True)
    mlp(xm, out=xm[:, :128], ln=(gam, bet), accumulate=True)
    assert torch.equal(xm[:, 128:], keep[:, 128:])
    got = xm[:, :128].double().cpu()
    assert ((got - want).abs() / (1 + want.abs())).max().item() < 6e-6, ((got - want).abs() / (1 + want.abs())).max().item()
    assert ((got - two[:, :128].double().cpu()).abs() / (1 + want.abs())).max().item() < 6e-6
    # without the residual, into a separate buffer
    out = torch.full((M, 128), 3.0, device=DEV)
    mlp(keep, out=out, ln=(gam, bet))
    want0 = want - keep[:, :128].double().cpu()
    assert ((out.double().cpu() - want0).abs() / (1 + want0.abs())).max().item() < 6e-6
