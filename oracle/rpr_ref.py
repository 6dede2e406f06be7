"""TEST INFRASTRUCTURE ONLY (never imported by the product package): restatement of the regression model's aggregator
with the correlation volume materialised -- the checker for csrc/corr_warp.hip -- and the deterministic weight filler the
golden generator (oracle/gen_rpr_golden.py, which EXECUTES the reference's own lib/models/regression code) and the tests
share, so that fixtures need not store 60 MB of weights.

Follows lib/models/regression/aggregator.py:42-116 (CorrelationVolumeWarping.forward) and :134-191 (QKV variant).
Pinned by tests/golden/ref_rpr_*.npz: outputs AND gradients of the reference's own modules on seeded inputs
(tests/test_rpr_oracle.py)."""
import zlib

import torch


def fill_deterministic(module, seed=0):
    """overwrite every parameter and buffer with values that depend only on (its state-dict key, its shape, seed)"""
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(zlib.crc32(f"{seed}:{name}".encode()))
            r = torch.randn(t.shape, generator=g)
            if name.endswith("running_var"):
                v = 1.0 + 0.2 * r.abs()
            elif name.endswith("running_mean"):
                v = 0.1 * r
            elif t.dim() >= 2:                                    # conv / linear weight: He scaling keeps activations O(1)
                fan_in = t[0].numel()
                v = r * (1.6 / fan_in) ** 0.5
            elif name.endswith("weight"):                         # norm scale
                v = 1.0 + 0.1 * r
            elif name.endswith("bin_score"):
                v = torch.full(t.shape, 2.0)
            else:                                                 # biases, s_r / s_t
                v = 0.05 * r
            t.copy_(v.to(t.dtype))
    return module


def position_grid(H, W, dtype=torch.float32, device="cpu"):
    u = torch.linspace(-1, 1, H, device=device).to(dtype)
    v = torch.linspace(-1, 1, W, device=device).to(dtype)
    return torch.stack([u[:, None].expand(H, W), v[None, :].expand(H, W)], 0).reshape(2, H * W)


def corr_warp_materialised(q, k, v, grid=None):
    """softmax(q^T k, dim=2) written out: warped = v c^T, pos = grid c^T, max = max_j c.  Any dtype / device; autograd-able."""
    c = torch.softmax(torch.bmm(q.transpose(1, 2), k), dim=2)
    warped = torch.bmm(v, c.transpose(1, 2))
    pos = torch.matmul(grid.to(c.dtype), c.transpose(1, 2)) if grid is not None else None
    return warped, pos, c.max(dim=2)[0]


def aggregate_materialised(vol0, vol1, position_encoder=True, max_score=True, half=False, normalise=False, im1=False):
    """CorrelationVolumeWarping.forward for the options the fused kernel serves (no dustbin / CV block)"""
    B, D, H, W = vol0.shape
    a, b = vol0.reshape(B, D, H * W), vol1.reshape(B, D, H * W)
    if normalise:
        a, b = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(b, dim=1)
    grid = position_grid(H, W, a.dtype, a.device) if position_encoder else None
    d = D // 2 if half else D
    warped, pos, mx = corr_warp_materialised(a[:, :d], b[:, :d], b, grid)
    out = [a, warped]
    if position_encoder:
        out.append(pos)
        if im1:
            out.append(grid[None].expand(B, -1, -1))
    if max_score:
        out.append(mx[:, None])
    return torch.cat(out, 1).reshape(B, -1, H, W)


class MaterialisedAggregator(torch.nn.Module):
    """drop-in for the product's CorrelationVolumeWarping in CPU tests of the rest of the model"""

    def __init__(self, product_aggregator):
        super().__init__()
        p = product_aggregator
        self.kw = dict(position_encoder=p.position_encoder, max_score=p.max_score_channel, half=p.cv_half_channels,
                       normalise=p.normalise_dot_prod, im1=p.position_encoder_im1)
        self.num_out_layers = p.num_out_layers

    def forward(self, vol0, vol1):
        return aggregate_materialised(vol0.float(), vol1.float(), **self.kw)
