"""-m gpu: the f16x2 RANGE GUARD (include/mfr_hip.h mfr_f16x2_guard_bind, csrc/guard.h, pipeline.RangeGuard).

The reference's networks are plain fp32 modules (etc/feature_matching_baselines/matchers.py:50,105): no |x| <= 65504 precondition.  The f16x2 kernels
have one, so (i) every f16x2 kernel must RAISE the bound device flag when a single operand element is out of range (or non-finite) and must not raise it
otherwise, wherever the element sits in the tile; (ii) a pipeline whose activations leave the range must hand out the exact (bf16x3) result, never inf /
NaN / a laundered number."""
import numpy as np
import pytest
import torch

from mapfree_reloc_amd import _lib, images as IM, options
from mapfree_reloc_amd.nets import weights as WT
from mapfree_reloc_amd.nets.conv import IgemmConv, WinoConv3x3
from mapfree_reloc_amd.nets.linear import SplitBatchedNT, SplitLinear
from mapfree_reloc_amd.pipeline import ST_RANGE, LoFTREmatPipeline, RangeGuard, SuperGluePnPPipeline, rerun_out_of_range

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def guard():
    options.reset()
    g = RangeGuard(torch.device(DEV), lambda: None)
    assert g.active
    return g


def _fires(guard, fn):
    with guard:
        y = fn()
    torch.cuda.synchronize()
    return int(guard.flag.item()) != 0, y


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (128, 128, 64), (1000, 512, 96), (77, 130, 32)])
def test_linear_layer_flags_one_out_of_range_element(guard, M, N, K):
    """K = 96 / 32 run the register-staged kernel (K % 64 != 0), the others the LDS-DMA default; one element of x at every corner of the problem"""
    g = torch.Generator().manual_seed(M + N + K)
    lin = SplitLinear((torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), torch.randn(N, generator=g).to(DEV))
    x = torch.randn(M, K, generator=g).to(DEV)
    fired, y = _fires(guard, lambda: lin(x, relu=True))
    assert not fired and torch.isfinite(y).all()
    for (r, c) in ((0, 0), (M - 1, K - 1), (M // 2, K // 3)):
        for bad in (7.0e4, -1.0e5, float("inf"), float("nan")):
            xb = x.clone(); xb[r, c] = bad
            fired, y = _fires(guard, lambda: lin(xb, relu=True))
            assert fired, (r, c, bad)
    xb = x.clone(); xb[3, 5] = 6.0e4                                   # the largest decade that IS in range
    fired, y = _fires(guard, lambda: lin(xb))
    assert not fired and torch.isfinite(y).all()
    # a ReLU would launder the NaN of the poisoned row (v_max_f32 drops a NaN operand): that is why the accumulators are tested, not the outputs
    xb = x.clone(); xb[0, 0] = 1.0e5
    with guard:
        y = lin(xb, relu=True)
    assert int(guard.flag.item()) == 1


def test_batched_score_product_flags(guard):
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(3, 200, 256, generator=g).to(DEV), torch.randn(3, 150, 256, generator=g).to(DEV)
    nt = SplitBatchedNT()
    fired, _ = _fires(guard, lambda: nt(a, b, out_mul=1 / 16))
    assert not fired
    a2 = a.clone(); a2[2, 199, 255] = 1e5
    fired, _ = _fires(guard, lambda: nt(a2, b, out_mul=1 / 16))
    assert fired


@pytest.mark.parametrize("ci,co,H,W,pool", [(64, 64, 48, 40, True), (128, 196, 30, 34, False), (196, 128, 17, 21, False)])
def test_winograd_layer_flags(guard, ci, co, H, W, pool):
    g = torch.Generator().manual_seed(ci + H)
    options.set("CONV_KERNEL", "split")
    try:
        conv = WinoConv3x3((torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(DEV), torch.randn(co, generator=g).to(DEV))
        x = torch.randn(2, ci, H, W, generator=g).to(DEV)
        fired, y = _fires(guard, lambda: conv(x, act=1, pool=pool))
        assert not fired and torch.isfinite(y).all()
        for pos in ((0, 0, 0, 0), (1, ci - 1, H - 1, W - 1), (1, ci // 2, H // 2, 1)):
            xb = x.clone(); xb[pos] = 7.0e4
            fired, _ = _fires(guard, lambda: conv(xb, act=1, pool=pool))
            assert fired, pos
        # the Winograd precondition is on the TRANSFORMED patch: four neighbours of 2e4 each are in range one by one, their combination is not
        xb = x.clone(); xb[0, 0, 4:8, 4:8] = 2.0e4; xb[0, 0, 5:7:1, 4:8] *= -1.0
        fired, _ = _fires(guard, lambda: conv(xb, act=1, pool=pool))
        assert fired
    finally:
        options.reset()


def test_igemm_convolution_flags(guard):
    g = torch.Generator().manual_seed(9)
    for (ci, co, k, s) in ((1, 128, 7, 2), (128, 196, 3, 2), (196, 256, 1, 1)):
        conv = IgemmConv((torch.randn(co, ci, k, k, generator=g) / (k * ci ** 0.5)).to(DEV), None, s)
        x = torch.randn(2, ci, 40, 36, generator=g).to(DEV)
        fired, y = _fires(guard, lambda: conv(x, relu=True))
        assert not fired and torch.isfinite(y).all()
        xb = x.clone(); xb[1, ci - 1, 38, 34] = -9.0e4               # (even coordinates: a stride-2 1x1 layer would not read an odd pixel)
        fired, _ = _fires(guard, lambda: conv(xb, relu=True))
        assert fired, (ci, co, k, s)


def test_attention_flags(guard):
    from mapfree_reloc_amd.nets.superglue import SuperGlueHIP
    sg = SuperGlueHIP(WT.superglue_state_dict(), DEV)
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(4, 1024, 768, generator=g).to(DEV)
    n_tok = torch.tensor([1024, 900, 700, 1024], dtype=torch.int32, device=DEV)
    for cross in (False, True):
        fired, y = _fires(guard, lambda: sg.attention(qkv, n_tok, cross))
        assert not fired and torch.isfinite(y).all()
        for col in (3, 256 + 70, 512 + 255):                          # an element of q, of k, of v
            bad = qkv.clone(); bad[2, 5, col] = 1.0e6                 # (q enters scaled by log2(e) / 8: its bound is 65504 / 0.18)
            fired, _ = _fires(guard, lambda: sg.attention(bad, n_tok, cross))
            assert fired, (cross, col)


def test_unbound_launches_do_not_touch_the_flag(guard):
    g = torch.Generator().manual_seed(1)
    lin = SplitLinear(torch.randn(128, 64, generator=g).to(DEV))
    x = torch.randn(64, 64, generator=g).to(DEV); x[0, 0] = 1e6
    guard.flag.zero_()
    lin(x)
    torch.cuda.synchronize()
    assert int(guard.flag.item()) == 0


def _scaled(sd, key, factor):
    sd = dict(sd)
    sd[key] = sd[key] * factor
    return sd


def test_loftr_pipeline_out_of_range_yields_the_bf16x3_result():
    """the first convolution's weights x 3e5: its outputs (the next layers' inputs) leave the f16x2 range.  The default pipeline must mark every pair of
    the batch ST_RANGE (NaN pose, no inf / laundered number) and rerun_out_of_range must return what an exact bf16x3 pipeline computes, bit for bit"""
    options.reset()
    sd = _scaled(WT.loftr_state_dict(), "backbone.conv1.weight", 3.0e5)
    sb = IM.synthetic_batch([5000, 5001])
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in sb.items() if isinstance(v, np.ndarray)}
    args = (d["images"], d["depth0"], d["depth1"], d["K0"], d["K1"], d["pair_ids"])
    pipe = LoFTREmatPipeline(DEV, loftr_state=sd)
    out = pipe(*args)
    assert (out["status"] == ST_RANGE).all() and torch.isnan(out["R"]).all()
    fixed = rerun_out_of_range(pipe, out, *args)
    assert pipe.guard.reruns == 1 and not (fixed["status"] == ST_RANGE).any()
    with options.override(SPLIT="bf16x3"):
        exact = LoFTREmatPipeline(DEV, loftr_state=sd)
        want = exact(*args)
    for k in ("status", "n_corr", "n_inliers"):
        assert torch.equal(fixed[k], want[k]), k
    ok = want["status"] == 0
    assert torch.equal(fixed["R"][ok], want["R"][ok]) and torch.equal(fixed["t"][ok], want["t"][ok])
    assert torch.isfinite(fixed["pts1"]).all()
    # and the in-range network is left alone: no flag, no twin
    pipe2 = LoFTREmatPipeline(DEV)
    o2 = pipe2(*args)
    assert not (o2["status"] == ST_RANGE).any() and rerun_out_of_range(pipe2, o2, *args) is o2 and pipe2.guard._twin is None


def test_superglue_pipeline_out_of_range_yields_the_bf16x3_result():
    options.reset()
    sp = _scaled(WT.superpoint_state_dict(), "conv1a.weight", 1.0e6)
    sb = IM.synthetic_batch([5000, 5001])
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in sb.items() if isinstance(v, np.ndarray)}
    args = (d["images"], d["depth0"], d["K0"], d["K1"], d["pair_ids"])
    pipe = SuperGluePnPPipeline(DEV, sp_state=sp)
    out = pipe(*args)
    assert (out["status"] == ST_RANGE).all()
    fixed = rerun_out_of_range(pipe, out, *args)
    with options.override(SPLIT="bf16x3"):
        want = SuperGluePnPPipeline(DEV, sp_state=sp)(*args)
    assert not (fixed["status"] == ST_RANGE).any()
    for k in ("status", "n_corr", "n_inliers", "pts0", "pts1"):
        assert torch.equal(fixed[k], want[k]), k
