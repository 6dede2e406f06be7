"""Aggregators of the regression model (reference: lib/models/regression/aggregator.py).

`CorrelationVolumeWarping` (aggregator.py:6-116) and `CorrelationVolumeWarpingQKV` (aggregator.py:119-191) run through
csrc/corr_warp.hip: softmax(q^T k) is consumed tile by tile inside the kernel, forward and backward; the [B, N, N]
correlation volume (1.57 GB fp32 at the shipped batch 10 / 6256 positions) is never written.  The kernel computes in
fp32 whatever the autocast dtype of the encoder, like autocast's own fp32 softmax.

Two rarely used options need the volume itself and keep a materialised device path (plain torch ops on the GPU, one
shipped config each): DUSTBIN (an extra learnable score column/row) and CV_OUTLAYERS > 0 (a conv over the volume).
There is no CPU path: tensors must live on the HIP device."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import check, load, ptr, stream_ptr
from .encoder import PreActBlock


class _CorrWarp(torch.autograd.Function):
    """(q, k, v, grid) -> (warped, pos, max_score); fp32, contiguous [B, D, N].  grid None -> pos has 0 channels."""

    @staticmethod
    def forward(ctx, q, k, v, grid):
        lib = load(require_gpu=True)
        if not (q.is_cuda and k.is_cuda and v.is_cuda):
            raise RuntimeError("correlation-volume warping runs on the HIP device only (no CPU fallback)")
        q, k, v = (t.detach().float().contiguous() for t in (q, k, v))
        B, Dq, N = q.shape
        if v.shape[1] != 32 or k.shape != q.shape or v.shape[2] != N:
            raise ValueError(f"corr_warp: q/k [B,{{16,32}},N] and v [B,32,N] expected, got {tuple(q.shape)} {tuple(k.shape)} {tuple(v.shape)}")
        g = None if grid is None else grid.detach().float().contiguous()
        warped = torch.empty_like(v)
        pos = torch.empty(B, 2 if g is not None else 0, N, device=v.device, dtype=torch.float32)
        stats = torch.empty(3, B, N, device=v.device, dtype=torch.float32)      # max_score, row_max, row_sum
        check(lib.mfr_corr_warp_fwd(ptr(q), ptr(k), ptr(v), ptr(g), B, Dq, N, ptr(warped), ptr(pos) if g is not None else None,
                                    ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), stream_ptr()), "mfr_corr_warp_fwd")
        ctx.save_for_backward(q, k, v, g, warped, pos, stats)
        return warped, pos, stats[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_warped, d_pos, d_max):
        q, k, v, g, warped, pos, stats = ctx.saved_tensors
        lib = load(require_gpu=True)
        B, Dq, N = q.shape
        d_warped = torch.zeros_like(warped) if d_warped is None else d_warped.float().contiguous()
        delta = (d_warped * warped).sum(1)
        if g is not None and d_pos is not None:
            d_pos = d_pos.float().contiguous()
            delta = delta + (d_pos * pos).sum(1)
        else:
            d_pos = None
        if d_max is not None:
            d_max = d_max.float().contiguous()
            delta = delta + d_max * stats[0]
        delta = delta.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(lib.mfr_corr_warp_bwd(ptr(q), ptr(k), ptr(v), ptr(g), B, Dq, N, ptr(d_warped), ptr(d_pos), ptr(d_max), ptr(delta),
                                    ptr(stats[1]), ptr(stats[2]), ptr(dq), ptr(dk), ptr(dv), stream_ptr()), "mfr_corr_warp_bwd")
        return dq, dk, dv, None


def _warp_small(q, k, v, grid):
    """the same three outputs by materialising the volume with torch ops on the device: for channel counts the kernel is
    not built for (the 1/16-resolution `ResNet` encoder: 1024 channels x ~350 positions, a 0.5 MB volume)"""
    c = torch.softmax(torch.bmm(q.transpose(1, 2), k), dim=2)
    pos = torch.matmul(grid, c.transpose(1, 2)) if grid is not None else q.new_zeros(q.shape[0], 0, q.shape[2])
    return torch.bmm(v, c.transpose(1, 2)), pos, c.max(dim=2)[0]


def corr_warp(q, k, v, grid=None):
    """softmax(q^T k) applied to v (and to `grid`), plus the row maximum; see csrc/corr_warp.hip.  Leaves autocast."""
    with torch.autocast("cuda", enabled=False):
        q, k, v = q.float(), k.float(), v.float()
        if v.shape[1] == 32 and q.shape[1] in (16, 32):
            return _CorrWarp.apply(q, k, v, grid)
        if not v.is_cuda:
            raise RuntimeError("correlation-volume warping runs on the HIP device only (no CPU fallback)")
        return _warp_small(q, k, v, grid)


def position_grid(H, W, device):
    """[2, H*W]: channel 0 = linspace(-1, 1, H) along rows, channel 1 = linspace(-1, 1, W) along columns (aggregator.py:80-83)"""
    u = torch.linspace(-1, 1, H, device=device)
    v = torch.linspace(-1, 1, W, device=device)
    return torch.stack([u[:, None].expand(H, W), v[None, :].expand(H, W)], 0).reshape(2, H * W).contiguous()


class CorrelationVolumeWarping(nn.Module):
    def __init__(self, cfg, volume_channels):
        super().__init__()
        self.position_encoder = bool(cfg.POSITION_ENCODER)
        self.position_encoder_im1 = bool(cfg.POSITION_ENCODER_IM1)
        self.max_score_channel = bool(cfg.MAX_SCORE_CHANNEL)
        self.cv_out_layers = int(cfg.CV_OUTLAYERS)
        self.cv_half_channels = bool(cfg.CV_HALF_CHANNELS)
        self.pos_encoder_channels = int(cfg.UPSAMPLE_POS_ENC)
        self.dustbin = bool(cfg.DUSTBIN)
        self.normalise_dot_prod = bool(cfg.NORMALISE_DOT)

        n_pos = (2 if self.position_encoder else 0) + (2 if self.position_encoder_im1 else 0)
        self.num_out_layers = 2 * volume_channels + n_pos + (1 if self.max_score_channel else 0)
        if self.cv_out_layers > 0:
            self.CV_block = PreActBlock(4800, self.cv_out_layers)        # 4800 = 80*60 positions (aggregator.py:27)
            self.num_out_layers += self.cv_out_layers
        if self.pos_encoder_channels > 0:
            self.pos_encoder_block = PreActBlock(n_pos, self.pos_encoder_channels)
            self.num_out_layers += self.pos_encoder_channels
        if self.dustbin:
            self.bin_score = nn.Parameter(100 * torch.ones(1, 1, 1))
            self.bin_feature = nn.Parameter(torch.zeros(1, volume_channels, 1), requires_grad=False)

    def _materialised(self, vol0, vol1, grid, H, W):
        """DUSTBIN / CV_OUTLAYERS: the options that need the volume itself (aggregator.py:62-66,105-110); device torch ops"""
        B, D, N = vol0.shape
        half = D // 2 if self.cv_half_channels else D
        vol0, vol1 = vol0.float(), vol1.float()
        with torch.autocast("cuda", enabled=False):
            c = torch.bmm(vol0[:, :half].transpose(1, 2), vol1[:, :half])
            feat = vol1
            if self.dustbin:
                c = F.pad(c, (0, 1, 0, 1))
                c[:, N, :] = self.bin_score.view(1, 1)
                c[:, :, N] = self.bin_score.view(1, 1)
                feat = torch.cat([vol1, self.bin_feature.float().expand(B, -1, -1)], 2)
            c = torch.softmax(c, dim=2)
            warped = torch.bmm(feat, c.transpose(1, 2))[:, :, :N]
            pos = torch.matmul(grid, c[:, :N, :N].transpose(1, 2)) if grid is not None else None
            mx = c.max(dim=2)[0][:, :N]
            reduced = None
            if self.cv_out_layers > 0:
                reduced = self.CV_block(c[:, :N, :N].reshape(B, N, H, W)).reshape(B, -1, N)
        return warped, pos, mx, reduced

    def forward(self, vol0, vol1):
        if vol0.shape != vol1.shape:
            raise ValueError("Feature volumes shape must match")
        B, D, H, W = vol0.shape
        N = H * W
        vol0 = vol0.reshape(B, D, N)
        vol1 = vol1.reshape(B, D, N)
        if self.normalise_dot_prod:
            vol0, vol1 = F.normalize(vol0, dim=1), F.normalize(vol1, dim=1)
        grid = position_grid(H, W, vol0.device) if self.position_encoder else None
        reduced = None
        if self.dustbin or self.cv_out_layers > 0:
            warped, pos, mx, reduced = self._materialised(vol0, vol1, grid, H, W)
        else:
            half = D // 2 if self.cv_half_channels else D
            q, k = (vol0[:, :half], vol1[:, :half]) if self.cv_half_channels else (vol0, vol1)
            warped, pos, mx = corr_warp(q, k, vol1, grid)
        out = [vol0.float(), warped]
        if self.position_encoder:
            out.append(pos)
            gridB = grid[None].expand(B, -1, -1)
            if self.position_encoder_im1:
                out.append(gridB)
            if self.pos_encoder_channels > 0:
                feats = torch.cat([pos, gridB], 1) if self.position_encoder_im1 else pos
                out.append(self.pos_encoder_block(feats.reshape(B, -1, H, W)).reshape(B, -1, N).float())
        if self.max_score_channel:
            out.append(mx[:, None])
        if reduced is not None:
            out.append(reduced.float())
        return torch.cat(out, dim=1).reshape(B, -1, H, W)


class CorrelationVolumeWarpingQKV(nn.Module):
    """learned 1x1 query / key / value projections (shared value projection for both images), then the same warping"""

    def __init__(self, cfg, volume_channels):
        super().__init__()
        self.position_encoder = bool(cfg.POSITION_ENCODER)
        self.max_score_channel = bool(cfg.MAX_SCORE_CHANNEL)
        self.normalise_dot_prod = bool(cfg.NORMALISE_DOT)
        self.residuals = bool(cfg.RESIDUAL_ATT)
        self.num_out_layers = 2 * volume_channels + (2 if self.position_encoder else 0) + (1 if self.max_score_channel else 0)
        self.Q_mlp = nn.Conv2d(volume_channels, volume_channels, 1, bias=False)
        self.K_mlp = nn.Conv2d(volume_channels, volume_channels, 1, bias=False)
        self.V_mlp = nn.Conv2d(volume_channels, volume_channels, 1, bias=False)

    def forward(self, vol0, vol1):
        if vol0.shape != vol1.shape:
            raise ValueError("Feature volumes shape must match")
        B, D, H, W = vol0.shape
        N = H * W
        q, k, v0, v1 = self.Q_mlp(vol0), self.K_mlp(vol1), self.V_mlp(vol0), self.V_mlp(vol1)
        if self.residuals:
            q, k, v0, v1 = q + vol0, k + vol1, v0 + vol0, v1 + vol1
        q, k, v0, v1 = (t.reshape(B, D, N) for t in (q, k, v0, v1))
        if self.normalise_dot_prod:
            q, k = F.normalize(q, p=2.0, dim=1), F.normalize(k, p=2.0, dim=1)
        grid = position_grid(H, W, vol0.device) if self.position_encoder else None
        warped, pos, mx = corr_warp(q, k, v1, grid)
        out = [v0.float(), warped]
        if self.position_encoder:
            out.append(pos)
        if self.max_score_channel:
            out.append(mx[:, None])
        return torch.cat(out, dim=1).reshape(B, -1, H, W)


class Concat(nn.Module):
    def __init__(self, cfg, volume_channels):
        super().__init__()
        self.num_out_layers = 2 * volume_channels

    def forward(self, vol0, vol1):
        return torch.cat([vol0, vol1], dim=1)


AGGREGATORS = {"CorrelationVolumeWarping": CorrelationVolumeWarping, "CorrelationVolumeWarpingQKV": CorrelationVolumeWarpingQKV,
               "Concat": Concat}
