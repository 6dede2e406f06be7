"""oracle/loftr_ref.py vs an INDEPENDENT implementation of the sub-steps LoFTR shares with EfficientLoFTR: HuggingFace
`transformers` (installed offline) restates kornia's `create_meshgrid` / `spatial_expectation2d`, LoFTR's `mask_border` and the
dual-softmax coarse matching (feature / sqrt(C), similarity / temperature, softmax over both axes, threshold, border removal,
mutual row / column maximum) in modeling_efficientloftr.py:778-1049.  The rest of EfficientLoFTR is a different
network, so only these steps can be cross-checked; they are fed the SAME tensors here.  CPU only."""
import types

import numpy as np
import pytest
import torch

from oracle import loftr_ref as LR

hf = pytest.importorskip("transformers.models.efficientloftr.modeling_efficientloftr")


def test_spatial_expectation2d_equals_hf():
    g = torch.Generator().manual_seed(0)
    for W in (5, 7):
        heat = torch.softmax(torch.randn(300, W * W, generator=g) * 3, 1).view(300, W, W)
        ours = LR.spatial_expectation2d(heat)
        theirs = hf.spatial_expectation2d(heat[None], normalized_coordinates=True)[0]      # [1, M, W, W] -> [1, M, 2]
        np.testing.assert_allclose(ours.numpy(), theirs.numpy(), rtol=0, atol=1e-6)
    # kornia's documented known answer (un-normalised grid): mass at (x = 1, y = 2)
    h = torch.zeros(1, 1, 3, 3); h[0, 0, 2, 1] = 1.0
    assert hf.spatial_expectation2d(h, False).tolist() == [[[1.0, 2.0]]]
    one = LR.spatial_expectation2d(h[0])                                                    # normalised: x = 0, y = +1
    np.testing.assert_allclose(one.numpy(), [[0.0, 1.0]], atol=1e-7)


def test_mask_border_equals_hf():
    g = torch.Generator().manual_seed(1)
    for b in (1, 2, 3):
        m = torch.rand(2, 9, 8, 10, 7, generator=g) > 0.3
        ours = m.clone(); LR.mask_border(ours, b, False)
        theirs = hf.mask_border(m.clone(), b, False)
        assert torch.equal(ours, theirs)
        assert not theirs[:, :b].any() and not theirs[:, :, :, :, -b:].any() and theirs[:, b:-b, b:-b, b:-b, b:-b].any()


def _hf_coarse(feat0, feat1, h, w, thr, border, temperature):
    """HF's _coarse_matching + _get_matches_from_scores, bound to a stand-in object carrying only the config they read"""
    cfg = types.SimpleNamespace(coarse_matching_threshold=thr, coarse_matching_border_removal=border,
                                coarse_matching_temperature=temperature, coarse_matching_skip_softmax=False)
    obj = types.SimpleNamespace(config=cfg)
    obj._get_matches_from_scores = types.MethodType(hf.EfficientLoFTRForKeypointMatching._get_matches_from_scores, obj)
    B, L, C = feat0.shape
    cf = torch.stack([feat0, feat1], 1).view(B, 2, h, w, C).permute(0, 1, 4, 2, 3)          # [B, 2, C, h, w]
    return hf.EfficientLoFTRForKeypointMatching._coarse_matching(obj, cf, 8.0)


@pytest.mark.parametrize("seed,thr", [(0, 0.2), (1, 0.2), (2, 0.05)])
def test_dual_softmax_coarse_matching_equals_hf(seed, thr):
    """same features -> same set of (i, j) matches, same confidences, same coarse keypoints (x, y) * 8"""
    g = torch.Generator().manual_seed(seed)
    B, h, w, C = 2, 12, 9, 64
    L = h * w
    f1 = torch.randn(B, L, C, generator=g) * 4.0
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(B)])
    f0 = torch.gather(f1, 1, perm[..., None].expand(-1, -1, C)) + 0.4 * torch.randn(B, L, C, generator=g)      # f0[i] ~ f1[perm[i]]
    f0[:, ::3] = torch.randn(B, (L + 2) // 3, C, generator=g) * 4.0                                            # a third without a partner
    ours = LR.coarse_matching(f0, f1, (h, w), (h, w), thr=thr, border_rm=2, temperature=0.1, scale=8)
    assert len(ours["b_ids"]) > 10
    for b in range(B):
        # one pair per call: HF concatenates the two index tensors along the BATCH axis before reshaping to [B, 2, L], which pairs
        # them correctly only for B = 1
        kp, sc, idx = (t[0] for t in _hf_coarse(f0[b:b + 1], f1[b:b + 1], h, w, thr, 2, 0.1))
        sel = ours["b_ids"] == b
        mine = {(int(i), int(j)): float(c) for i, j, c in zip(ours["i_ids"][sel], ours["j_ids"][sel], ours["mconf"][sel])}
        # HF: idx[b, 0, j] = the image-0 cell matched to cell j of image 1 (-1 = none); idx[b, 1, i] = the image-1 cell matched to i
        theirs = {(i, int(idx[1, i])): float(sc[1, i]) for i in range(L) if int(idx[1, i]) >= 0}
        theirs_rev = {(int(idx[0, j]), j) for j in range(L) if int(idx[0, j]) >= 0}
        assert set(mine) == set(theirs) == theirs_rev
        for k in mine:
            assert abs(mine[k] - theirs[k]) <= 1e-6 * max(1.0, theirs[k])
        # coarse keypoints: (i % w, i // w) * 8
        k0 = ours["mkpts0_c"][sel].numpy(); k1 = ours["mkpts1_c"][sel].numpy()
        for (i, j), a, c in zip(zip(ours["i_ids"][sel].tolist(), ours["j_ids"][sel].tolist()), k0, k1):
            np.testing.assert_array_equal(a, kp[0, j].numpy())      # HF slot j of row 0: the image-0 cell (x, y) * 8 matched to image-1 cell j
            np.testing.assert_array_equal(c, kp[1, i].numpy())      # HF slot i of row 1: the image-1 cell matched to image-0 cell i
