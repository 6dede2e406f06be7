"""Regression heads: residual units over the aggregated volume, then a small MLP to a pose parametrisation
(reference: lib/models/regression/head.py:10-323).  Class, attribute and `data` key names are the reference's.

The reference tests the outputs for NaN/Inf with four host round trips per step and calls sys.exit.  Here each head leaves
ONE device-side flag (`self.invalid`, no synchronisation); `RegressionModel.check_finite()` reads it -- immediately in
eval mode, every TRAINING.LOG_INTERVAL steps while training -- and raises SystemExit("Stopped"), the reference's exit."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoder import PreActBlock
from .geometry import procrustes, quaternion_to_rotation_matrix, rotation_matrix_from_ortho6d


def _mlp(out_dims, in_features=None):
    first = nn.Linear(in_features, 256) if in_features else nn.LazyLinear(256)
    return nn.Sequential(first, nn.ReLU(), nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, out_dims))


class _Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        # ONE persistent device flag, updated in place: a captured training step (TRAINING.GRAPH_STEP) bakes this buffer's address into
        # the graph, so every replay ORs into the flag the reader looks at and the reader's reset is seen by the next replay
        # (non-persistent: the reference's state_dict keys stay as they are)
        self.register_buffer("invalid", torch.zeros((), dtype=torch.bool), persistent=False)

    def _flag(self, *tensors):
        ok = torch.stack([torch.isfinite(t.detach()).all() for t in tensors]).all()
        # sticky on the device: a NaN in ANY forward since the last read stays visible (the reader clears it: clear_invalid)
        self.invalid.logical_or_(~ok)

    def clear_invalid(self):
        self.invalid.zero_()


class ResBlockMLP(_Trunk):
    """two stride-2 residual units, flattened (head.py:10-25)"""

    def __init__(self, cfg, in_channels):
        super().__init__()
        self.resblock1 = PreActBlock(in_channels, 256, stride=2)
        self.resblock2 = PreActBlock(256, 128, stride=2)

    def features(self, volume):
        return self.resblock2(self.resblock1(volume)).flatten(1)


class DeepResBlock(_Trunk):
    """four stride-2 residual units 64-128-256-512, global average pool if HEAD.AVG_POOL (head.py:28-52)"""

    def __init__(self, cfg, in_channels):
        super().__init__()
        bn = bool(cfg.HEAD.BATCH_NORM)
        self.avg_pool = bool(cfg.HEAD.AVG_POOL)
        widths = [in_channels, 64, 128, 256, 512]
        for i in range(4):
            setattr(self, f"resblock{i + 1}", PreActBlock(widths[i], widths[i + 1], stride=2, bn=bn))

    def features(self, volume):
        x = volume
        for i in range(1, 5):
            x = getattr(self, f"resblock{i}")(x)
        if self.avg_pool:
            x = x.mean(dim=(2, 3))
        return x.flatten(1)

    @property
    def feature_dim(self):
        return 512 if self.avg_pool else None


class _ProcrustesMixin:
    """3-D/3-D correspondences from the MLP output, solved by Kabsch (head.py:55-103 / 106-163)"""

    def _init_points(self, cfg):
        self.add_basis = bool(cfg.HEAD.ADD_BASIS)
        self.num_pts = int(cfg.HEAD.NUM_PTS)
        if not (self.num_pts == 3 or (self.num_pts % 2 == 0 and self.num_pts >= 6)):
            raise AssertionError("num_pts must be 3, 6 or a multiple of 2 higher than 6")

    def _solve(self, raw):
        B = raw.shape[0]
        xyz = raw.float().view(B, -1, 3)
        basis = torch.eye(3, device=xyz.device).expand(B, 3, 3)
        if self.num_pts == 3:
            cor0, cor1 = basis, xyz
        else:
            cor0, cor1 = xyz[:, :self.num_pts // 2], xyz[:, self.num_pts // 2:]
        if self.add_basis:
            if self.num_pts == 6:
                cor0 = cor0 + basis
            if self.num_pts in (3, 6):
                cor1 = cor1 + basis
        with torch.autocast(xyz.device.type, enabled=False):
            R, t = procrustes(cor0, cor1)
        self.xyz, self.R, self.t = xyz.detach(), R.detach(), t.detach()
        self._flag(xyz, R, t)
        return R, t


class ProcrustesResBlockMLP(ResBlockMLP, _ProcrustesMixin):
    def __init__(self, cfg, in_channels):
        super().__init__(cfg, in_channels)
        self._init_points(cfg)
        self.mlp = nn.LazyLinear(3 * self.num_pts)

    def forward(self, feature_volume, data):
        return self._solve(self.mlp(self.features(feature_volume)))


class ProcrustesDeepResBlock(DeepResBlock, _ProcrustesMixin):
    def __init__(self, cfg, in_channels):
        super().__init__(cfg, in_channels)
        self._init_points(cfg)
        self.mlp = _mlp(3 * self.num_pts, self.feature_dim)

    def forward(self, feature_volume, data):
        return self._solve(self.mlp(self.features(feature_volume)))


class QuatDeepResBlock(DeepResBlock):
    """quaternion + (unit direction, scale) or scaled translation (head.py:166-214); leaves q / t_direction / scale in data"""

    def __init__(self, cfg, in_channels):
        super().__init__(cfg, in_channels)
        self.regress_scale = bool(cfg.HEAD.SEPARATE_SCALE)
        self.output_dims = 8 if self.regress_scale else 7
        self.mlp = _mlp(self.output_dims, self.feature_dim)

    def forward(self, feature_volume, data):
        x = self.mlp(self.features(feature_volume)).float()
        B = x.shape[0]
        quat = F.normalize(x[:, :4], dim=1)
        data["q"] = quat
        R = quaternion_to_rotation_matrix(quat)
        if self.regress_scale:
            scale = x[:, 4].abs().view(B, 1, 1)
            direction = F.normalize(x[:, 5:], dim=1).view(B, 1, 3)
            t = scale * direction
            data["t_direction"], data["scale"] = direction, scale
        else:
            t = x[:, 4:].view(B, 1, 3)
        self._flag(R, t)
        return R, t


class _DirectMixin:
    """6-D rotation + translation (head.py:217-270)"""

    def _direct(self, out):
        out = out.float().view(-1, 9)
        return rotation_matrix_from_ortho6d(out[:, :6]), out[:, 6:].view(-1, 1, 3)


class DirectResBlockMLP(ResBlockMLP, _DirectMixin):
    def __init__(self, cfg, in_channels):
        super().__init__(cfg, in_channels)
        self.mlp = nn.LazyLinear(3 + 6)

    def forward(self, feature_volume, data):
        return self._direct(self.mlp(self.features(feature_volume)))


class DirectDeepResBlockMLP(DeepResBlock, _DirectMixin):
    def __init__(self, cfg, in_channels):
        super().__init__(cfg, in_channels)
        self.mlp = _mlp(3 + 6, self.feature_dim)

    def forward(self, feature_volume, data):
        return self._direct(self.mlp(self.features(feature_volume)))


def _euler_xyz_deg_to_matrix(angles):
    """extrinsic x-y-z Euler angles in degrees [b, 3] -> R = Rz Ry Rx (scipy Rotation.from_euler('xyz'), head.py:303-306)"""
    a = torch.deg2rad(angles.double())
    cx, cy, cz = torch.cos(a).unbind(1)
    sx, sy, sz = torch.sin(a).unbind(1)
    rows = [cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
            sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
            -sy, cy * sx, cy * cx]
    return torch.stack(rows, 1).view(-1, 3, 3).float()


class AngularBinsDeepResBlockMLP(DeepResBlock):
    """rotation as three 1-degree classification problems (360 / 180 / 360 bins), translation direct or as spherical bins
    + scale (head.py:273-323).  The arg-max -> rotation step stays on the device (the reference round-trips through scipy)."""

    def __init__(self, cfg, in_channels):
        super().__init__(cfg, in_channels)
        self.regress_scale_separately = bool(cfg.HEAD.SEPARATE_SCALE)
        out = 900 + (360 + 180 + 1 if self.regress_scale_separately else 3)
        self.mlp = nn.Linear(self.feature_dim, out) if self.feature_dim else nn.LazyLinear(out)

    def forward(self, feature_volume, data):
        out = self.mlp(self.features(feature_volume)).float()
        B = out.shape[0]
        R_bins = out[:, :900]
        data["R_bins"] = R_bins
        with torch.no_grad():
            idx = torch.stack([R_bins[:, :360].argmax(1), R_bins[:, 360:540].argmax(1), R_bins[:, 540:].argmax(1)], 1)
            R = _euler_xyz_deg_to_matrix(idx - torch.tensor([[180, 90, 180]], device=out.device))
        if self.regress_scale_separately:
            phi_bins, theta_bins = out[:, 900:1260], out[:, 1260:1440]
            scale = out[:, -1:].abs()
            data["t_sph_phi"], data["t_sph_theta"], data["scale"] = phi_bins, theta_bins, scale.view(B, 1, 1)
            phi = torch.deg2rad(phi_bins.argmax(1).float()).view(B, 1)
            theta = torch.deg2rad(theta_bins.argmax(1).float()).view(B, 1)
            t = scale * torch.cat([torch.cos(phi) * torch.sin(theta), torch.sin(phi) * torch.sin(theta), torch.cos(theta)], 1)
        else:
            t = out[:, 900:]
        return R, t.view(B, 1, 3)


HEADS = {c.__name__: c for c in (ProcrustesResBlockMLP, ProcrustesDeepResBlock, QuatDeepResBlock, DirectResBlockMLP,
                                 DirectDeepResBlockMLP, AngularBinsDeepResBlockMLP)}
