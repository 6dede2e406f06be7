"""Cross-check of the oracle's SuperPoint / SuperGlue restatement (oracle/nets_ref.py) against the
independent HuggingFace `transformers` implementations with IDENTICAL (seeded random) weights.
Neither is the reference's own dependency (magicleap submodule, empty offline) -- this pins our
restatement to a second, third-party statement of the same published algorithms."""
import numpy as np
import pytest
import torch

from oracle import nets_ref as NR
from mapfree_reloc_amd.nets import weights as WT

tr = pytest.importorskip("transformers")


def _image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(1, 1, h // 8, w // 8, generator=g)
    img = torch.nn.functional.interpolate(img, size=(h, w), mode="bicubic", align_corners=False)
    img = img + 0.15 * torch.rand(1, 1, h, w, generator=g)
    return img.clamp(0, 1)


def _hf_superpoint(sp, max_kp):
    from transformers import SuperPointConfig, SuperPointForKeypointDetection
    cfg = SuperPointConfig(keypoint_threshold=0.005, max_keypoints=max_kp, nms_radius=4, border_removal_distance=4)
    hf = SuperPointForKeypointDetection(cfg).eval()
    sd = sp.state_dict()
    m = {}
    for i, blk in enumerate(["1", "2", "3", "4"]):
        for ab in "ab":
            for wb in ("weight", "bias"):
                m[f"encoder.conv_blocks.{i}.conv_{ab}.{wb}"] = sd[f"conv{blk}{ab}.{wb}"]
    for wb in ("weight", "bias"):
        m[f"keypoint_decoder.conv_score_a.{wb}"] = sd[f"convPa.{wb}"]
        m[f"keypoint_decoder.conv_score_b.{wb}"] = sd[f"convPb.{wb}"]
        m[f"descriptor_decoder.conv_descriptor_a.{wb}"] = sd[f"convDa.{wb}"]
        m[f"descriptor_decoder.conv_descriptor_b.{wb}"] = sd[f"convDb.{wb}"]
    missing, unexpected = hf.load_state_dict(m, strict=True)
    return hf


def test_superpoint_ref_matches_hf():
    """all keypoints above threshold (no top-k), raster order.  HF 5.15 applies the upper border
    test against (8H, 8W) instead of (H, W) (modeling_superpoint.py:243-245 passes the already
    full-resolution size times 8), so its extra right/bottom-border keypoints are dropped here
    before comparing; upstream SuperPoint removes both borders."""
    sp = NR.SuperPointRef(max_keypoints=-1).eval()
    sp.load_state_dict(WT.superpoint_state_dict(1234))
    hf = _hf_superpoint(sp, -1)
    H, W = 240, 184
    img = _image(H, W, 3)
    (kp, sc, desc), = sp(img)
    with torch.no_grad():
        out = hf(img.expand(1, 3, H, W))
    n = int(out.mask[0].sum())
    hk = (out.keypoints[0, :n] * torch.tensor([W, H])).numpy()
    hs, hd = out.scores[0, :n].numpy(), out.descriptors[0, :n].numpy()
    keep = (hk[:, 0] < W - 4 - 0.5) & (hk[:, 1] < H - 4 - 0.5)
    hk, hs, hd = hk[keep], hs[keep], hd[keep]
    assert len(kp) == len(hk) and len(kp) > 200
    np.testing.assert_allclose(hk, kp.numpy(), atol=1e-3)
    np.testing.assert_allclose(hs, sc.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(hd, desc.t().numpy(), atol=2e-6)


def test_superpoint_ref_topk_order():
    sp = NR.SuperPointRef(max_keypoints=-1).eval()
    sp.load_state_dict(WT.superpoint_state_dict(1234))
    img = _image(240, 184, 4)
    (kp_all, sc_all, _), = sp(img)
    sp.max_keypoints = 100
    (kp, sc, _), = sp(img)
    assert len(kp) == 100
    order = np.lexsort((np.arange(len(sc_all)), -sc_all.numpy()))[:100]     # score desc, raster asc
    np.testing.assert_array_equal(kp.numpy(), kp_all.numpy()[order])
    np.testing.assert_array_equal(sc.numpy(), sc_all.numpy()[order])


def _hf_superglue(sg):
    from transformers import SuperGlueConfig, SuperGlueForKeypointMatching
    cfg = SuperGlueConfig(sinkhorn_iterations=20, matching_threshold=0.2)
    hf = SuperGlueForKeypointMatching(cfg).eval()
    sd = sg.state_dict()
    perm = torch.tensor([(c % 64) * 4 + c // 64 for c in range(256)])        # hf channel -> magicleap channel
    hsd = hf.state_dict()
    new = {}

    def lin(w):
        return w.squeeze(-1)

    def put_bn(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            new[f"{dst}.{k}"] = sd[f"{src}.{k}"]
    for i in range(4):
        new[f"keypoint_encoder.encoder.{i}.linear.weight"] = lin(sd[f"kenc.encoder.{3 * i}.weight"])
        new[f"keypoint_encoder.encoder.{i}.linear.bias"] = sd[f"kenc.encoder.{3 * i}.bias"]
        put_bn(f"keypoint_encoder.encoder.{i}.batch_norm", f"kenc.encoder.{3 * i + 1}")
    new["keypoint_encoder.encoder.4.weight"] = lin(sd["kenc.encoder.12.weight"])
    new["keypoint_encoder.encoder.4.bias"] = sd["kenc.encoder.12.bias"]
    for l in range(18):
        for j, nm in enumerate(["query", "key", "value"]):
            new[f"gnn.layers.{l}.attention.self.{nm}.weight"] = lin(sd[f"gnn.layers.{l}.attn.proj.{j}.weight"])[perm]
            new[f"gnn.layers.{l}.attention.self.{nm}.bias"] = sd[f"gnn.layers.{l}.attn.proj.{j}.bias"][perm]
        new[f"gnn.layers.{l}.attention.output.dense.weight"] = lin(sd[f"gnn.layers.{l}.attn.merge.weight"])[:, perm]
        new[f"gnn.layers.{l}.attention.output.dense.bias"] = sd[f"gnn.layers.{l}.attn.merge.bias"]
        new[f"gnn.layers.{l}.mlp.0.linear.weight"] = lin(sd[f"gnn.layers.{l}.mlp.0.weight"])
        new[f"gnn.layers.{l}.mlp.0.linear.bias"] = sd[f"gnn.layers.{l}.mlp.0.bias"]
        put_bn(f"gnn.layers.{l}.mlp.0.batch_norm", f"gnn.layers.{l}.mlp.1")
        new[f"gnn.layers.{l}.mlp.1.weight"] = lin(sd[f"gnn.layers.{l}.mlp.3.weight"])
        new[f"gnn.layers.{l}.mlp.1.bias"] = sd[f"gnn.layers.{l}.mlp.3.bias"]
    new["final_projection.final_proj.weight"] = lin(sd["final_proj.weight"])
    new["final_projection.final_proj.bias"] = sd["final_proj.bias"]
    new["bin_score"] = sd["bin_score"].reshape(hsd["bin_score"].shape)
    for k in hsd:
        if k.startswith("keypoint_detector."):
            new[k] = hsd[k]
    hf.load_state_dict(new, strict=True)
    return hf


def test_superglue_ref_matches_hf():
    sg = NR.SuperGlueRef().eval()
    sg.load_state_dict(WT.superglue_state_dict(99))   # structured-random: matches exist
    hf = _hf_superglue(sg)
    g = torch.Generator().manual_seed(5)
    N, H, W = 160, 240, 184
    k0 = torch.rand(1, N, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    perm = torch.randperm(N, generator=g)
    k1 = k0[:, perm] + torch.randn(1, N, 2, generator=g)
    s0 = torch.rand(1, N, generator=g); s1 = s0[:, perm]
    d0 = torch.nn.functional.normalize(torch.randn(1, 256, N, generator=g), dim=1)
    d1 = torch.nn.functional.normalize(d0[:, :, perm] + 0.1 * torch.randn(1, 256, N, generator=g), dim=1)
    pred = sg(k0, s0, d0, k1, s1, d1, (H, W))
    with torch.no_grad():
        kp = torch.stack([k0, k1], 1)                         # [1,2,N,2]
        ds = torch.stack([d0.transpose(1, 2), d1.transpose(1, 2)], 1)
        sc = torch.stack([s0, s1], 1)
        matches, mscores, _, _ = hf._match_image_pair(kp, ds, sc, H, W, mask=torch.ones(1, 2, N, dtype=torch.int))
    m0 = pred["matches0"][0].numpy()
    assert (m0 > -1).sum() > N // 4, "test needs real matches"
    np.testing.assert_array_equal(matches[0, 0].numpy(), m0)
    np.testing.assert_array_equal(matches[0, 1].numpy(), pred["matches1"][0].numpy())
    np.testing.assert_allclose(mscores[0, 0].numpy(), pred["matching_scores0"][0].numpy(), rtol=2e-4, atol=1e-6)
