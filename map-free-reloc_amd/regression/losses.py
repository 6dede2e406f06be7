"""Training losses and validation metrics of the regression model.
Reference: lib/utils/loss.py:10-240 (every loss takes the shared `data` dict: predictions `R`, `t` (+ head extras) and
ground truth `T_0to1`), lib/utils/metrics.py:6-47,50-70,119-133 (pose_error_torch, error_auc, A_metrics)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .geometry import rotation_matrix_to_quaternion


def _gt(data):
    T = data["T_0to1"]
    return T[:, :3, :3], T[:, :3, 3:].transpose(1, 2)          # R [b,3,3], t [b,1,3]


def _residual_rotation_loss(data, criterion):
    Rgt, _ = _gt(data)
    R = data["R"]
    eye = torch.eye(3, device=R.device, dtype=R.dtype).expand_as(R)
    return criterion(Rgt.transpose(1, 2) @ R, eye)


def rot_frobenius_loss(data):
    return _residual_rotation_loss(data, F.mse_loss)


def rot_l1_loss(data):
    return _residual_rotation_loss(data, F.l1_loss)


def _acos_clipped(c):
    return torch.acos(c.clamp(-0.99999, 0.99999))               # keeps the gradient finite at 0 and pi


def rot_angle_loss(data):
    """mean residual rotation angle [rad]"""
    Rgt, _ = _gt(data)
    tr = torch.diagonal(data["R"].transpose(1, 2) @ Rgt, dim1=-2, dim2=-1).sum(-1)
    return _acos_clipped((tr - 1) / 2).abs().mean()


def _euler_xyz_deg(R):
    """extrinsic x-y-z Euler angles [deg] of rotation matrices (scipy as_euler('xyz'), loss.py:46-49), gimbal-lock safe enough
    for bin targets: R = Rz(c) Ry(b) Rx(a)"""
    b = torch.asin((-R[:, 2, 0]).clamp(-1, 1))
    a = torch.atan2(R[:, 2, 1], R[:, 2, 2])
    c = torch.atan2(R[:, 1, 0], R[:, 0, 0])
    return torch.rad2deg(torch.stack([a, b, c], 1))


def rot_bin_loss(data):
    Rgt, _ = _gt(data)
    bins = data["R_bins"]
    target = torch.round(_euler_xyz_deg(Rgt.double()) + torch.tensor([[180., 90., 180.]], device=bins.device, dtype=torch.float64)).long()
    hi = torch.tensor([359, 179, 359], device=bins.device)
    target = torch.minimum(target.clamp_min(0), hi)
    parts = (bins[:, :360], bins[:, 360:540], bins[:, 540:])
    return sum(F.cross_entropy(p, target[:, i]) for i, p in enumerate(parts)) / 3


def trans_l2_loss(data):
    return F.mse_loss(data["t"], _gt(data)[1])


def trans_l1_loss(data):
    return F.l1_loss(data["t"], _gt(data)[1])


def _quat_gt(data):
    q = rotation_matrix_to_quaternion(_gt(data)[0].contiguous())
    return q * torch.sign(q[:, 0:1])                            # one hemisphere: q and -q are the same rotation


def quat_l1_loss(data):
    return F.l1_loss(data["q"], _quat_gt(data))


def robust_quat_l1_loss(data):
    """Hartley et al. quaternion distance min(|q - q*|, |q + q*|)"""
    q, qgt = data["q"], _quat_gt(data)
    return torch.minimum((q + qgt).norm(dim=1), (q - qgt).norm(dim=1)).mean()


def trans_scale_direction_loss(data):
    tgt = _gt(data)[1]
    return F.l1_loss(data["scale"], tgt.norm(dim=-1, keepdim=True)) + F.l1_loss(data["t_direction"], F.normalize(tgt, dim=-1))


def trans_ang_loss(data):
    t, tgt = data["t"], _gt(data)[1]
    cos = (t * tgt).sum(-1) / (t.norm(dim=-1) * tgt.norm(dim=-1) + 1e-6)
    ang = _acos_clipped(cos)
    return torch.minimum(ang, math.pi - ang).abs().mean()


def trans_sphbin_loss(data):
    tgt = _gt(data)[1]
    d = F.normalize(tgt, dim=-1).reshape(-1, 3)
    theta = torch.acos(d[:, 2])
    phi = torch.atan2(d[:, 1], d[:, 0] + 1e-5)
    phi = torch.where(phi < 0, phi + 2 * math.pi, phi)
    theta_bin = torch.round(torch.rad2deg(theta)).long().clamp(0, 179)
    phi_bin = torch.round(torch.rad2deg(phi)).long()
    phi_bin = torch.where(phi_bin == 360, torch.zeros_like(phi_bin), phi_bin)
    scale_l = F.l1_loss(data["scale"], tgt.norm(dim=-1, keepdim=True))
    return scale_l + (F.cross_entropy(data["t_sph_phi"], phi_bin) + F.cross_entropy(data["t_sph_theta"], theta_bin)) / 2


def trans_scale_l1_loss(data):
    return F.l1_loss(data["scale"], _gt(data)[1].norm(dim=-1, keepdim=True))


def empty_loss(data):
    return torch.zeros(1, device=data["T_0to1"].device, dtype=torch.float32)


LOSSES = {f.__name__: f for f in (rot_frobenius_loss, rot_l1_loss, rot_angle_loss, rot_bin_loss, trans_l2_loss, trans_l1_loss,
                                  quat_l1_loss, robust_quat_l1_loss, trans_scale_direction_loss, trans_ang_loss,
                                  trans_sphbin_loss, trans_scale_l1_loss, empty_loss)}


# ---- validation metrics --------------------------------------------------------------------------------------------
def pose_error_torch(R, t, Tgt, reduce=None):
    """per-sample translation angle / scale ratio / Euclidean error and rotation angle [deg, -, m, deg]"""
    Rgt, tgt = Tgt[:, :3, :3], Tgt[:, :3, 3:].transpose(1, 2)
    nt, ng = t.norm(dim=-1), tgt.norm(dim=-1)
    ang = torch.rad2deg(torch.acos(((t * tgt).sum(-1) / (nt * ng + 1e-9)).clamp(-1, 1)))
    ang = torch.minimum(ang, 180 - ang)
    tr = torch.diagonal(R.transpose(1, 2) @ Rgt, dim1=-2, dim2=-1).sum(-1)
    out = {"t_err_ang": ang, "t_err_scale": nt / ng, "t_err_scale_sym": torch.maximum(nt / ng, ng / nt),
           "t_err_euc": (t - tgt).norm(dim=-1), "R_err": torch.rad2deg(torch.acos(((tr - 1) / 2).clamp(-1, 1)))}
    if reduce is not None:
        fn = {"mean": torch.mean, "median": torch.median}[reduce]
        for k in ("t_err_ang", "t_err_scale", "t_err_euc", "R_err"):
            out[k] = fn(out[k])
    return out


def error_auc(errors, thresholds):
    """area under the recall-vs-error curve up to each threshold, normalised (NaN counts as a miss)"""
    e = np.sort(np.nan_to_num(np.asarray(errors, dtype=np.float64), nan=np.inf))
    e = np.concatenate([[0.0], e])
    recall = np.linspace(0, 1, len(e))
    out = {}
    for thr in thresholds:
        last = int(np.searchsorted(e, thr))
        x = np.concatenate([e[:last], [thr]])
        y = np.concatenate([recall[:last], [recall[last - 1]]])
        out[f"auc@{thr}"] = float(np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]) / 2) / thr)
    return out


def A_metrics(t_scale_err_sym):
    s = torch.as_tensor(t_scale_err_sym)
    return tuple((s < 1.25 ** k).float().mean() for k in (1, 2, 3))
