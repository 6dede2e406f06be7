"""Small differentiable rotation / registration helpers of the regression heads and losses.
Reference: lib/utils/solver.py:4-37 (batched Kabsch `procrustes`), lib/utils/rotationutils.py:10-55 (6-D rotation),
kornia.geometry.conversions.{quaternion_to_rotation_matrix, rotation_matrix_to_quaternion} as called from
lib/models/regression/head.py:187 and lib/utils/loss.py:28-32 (kornia itself is an un-vendored dependency; w-first order)."""
import torch
import torch.nn.functional as F


class _KabschRotation(torch.autograd.Function):
    """H [b,3,3] (= A_c^T B_c) -> the reflection-corrected Kabsch rotation, forward and backward through csrc/kabsch.hip: Horn's
    quaternion form + Jacobi sweeps instead of an SVD, the gradient in closed form through the polar factor.  Same rotation and
    same gradient as the SVD route (tests/test_kabsch_host.py: 1e-13 in binary64), but with no host synchronisation --
    torch.linalg.svd reads the solver status back, which stalls the launch queue every step and forbids HIP-graph capture."""

    @staticmethod
    def forward(ctx, H):
        from .._lib import check, load, ptr, stream_ptr
        lib = load(require_gpu=True)
        H = H.detach().float().contiguous()
        R = torch.empty_like(H)
        check(lib.mfr_kabsch_fwd(ptr(H), H.shape[0], ptr(R), stream_ptr()), "mfr_kabsch_fwd")
        ctx.save_for_backward(H)
        return R

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gR):
        from .._lib import check, load, ptr, stream_ptr
        (H,) = ctx.saved_tensors
        gR = gR.float().contiguous()
        gH = torch.empty_like(H)
        check(load(require_gpu=True).mfr_kabsch_bwd(ptr(H), ptr(gR), H.shape[0], ptr(gH), stream_ptr()), "mfr_kabsch_bwd")
        return gH


SYNC_FREE_KABSCH = True        # device tensors: csrc/kabsch.hip; False: torch.linalg.svd everywhere (A/B switch)


def procrustes(A, B):
    """least-squares rigid registration of point sets with known correspondences (Kabsch): R, t with B ~ A R^T + t.
    A, B [b, n, 3] -> R [b, 3, 3], t [b, 1, 3].  fp32 on the device (kernel), fp32/fp64 on the host (SVD)."""
    if A.dim() != 3 or B.dim() != 3:
        raise AssertionError("three dimensions are required")
    if A.shape != B.shape:
        raise AssertionError("batch size, number of correspondences and spatial dimensions must match")
    if A.shape[2] != 3:
        raise AssertionError("number of spatial dimensions must be 3")
    a0, b0 = A.mean(dim=1, keepdim=True), B.mean(dim=1, keepdim=True)
    H = (A - a0).transpose(1, 2) @ (B - b0)
    if SYNC_FREE_KABSCH and H.is_cuda and H.dtype == torch.float32:
        R = _KabschRotation.apply(H)
        return R, b0 - a0 @ R.transpose(1, 2)
    U, _, Vh = torch.linalg.svd(H)
    V = Vh.transpose(1, 2)
    # proper rotation: flip the last singular direction when det(V U^T) < 0
    flip = torch.ones(A.shape[0], 3, dtype=A.dtype, device=A.device)
    flip[:, 2] = torch.sign(torch.linalg.det(U @ Vh))
    R = (V * flip[:, None, :]) @ U.transpose(1, 2)
    return R, b0 - a0 @ R.transpose(1, 2)


def _unit(v, floor=1e-8):
    return v / v.norm(dim=1, keepdim=True).clamp_min(floor)


def rotation_matrix_from_ortho6d(poses):
    """Zhou et al. continuous 6-D parametrisation: columns x, y, z from two raw 3-vectors (Gram-Schmidt via cross products)"""
    x = _unit(poses[:, 0:3])
    z = _unit(torch.linalg.cross(x, poses[:, 3:6], dim=1))
    y = torch.linalg.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


def quaternion_to_rotation_matrix(q):
    """unit quaternion (w, x, y, z) [b, 4] -> rotation matrix [b, 3, 3]"""
    q = F.normalize(q, dim=1)
    w, x, y, z = q.unbind(1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    rows = [1 - (ty * y + tz * z), tx * y - tz * w, tx * z + ty * w,
            tx * y + tz * w, 1 - (tx * x + tz * z), ty * z - tx * w,
            tx * z - ty * w, ty * z + tx * w, 1 - (tx * x + ty * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def rotation_matrix_to_quaternion(R, eps=1e-8):
    """rotation matrix [b, 3, 3] -> quaternion (w, x, y, z); branch on the largest of (trace, R00, R11, R22) for stability"""
    m = R.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(1)
    tr = m00 + m11 + m22
    cand = []
    s = torch.sqrt((tr + 1).clamp_min(eps)) * 2
    cand.append(torch.stack([0.25 * s, (m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s], 1))
    s = torch.sqrt((1 + m00 - m11 - m22).clamp_min(eps)) * 2
    cand.append(torch.stack([(m21 - m12) / s, 0.25 * s, (m01 + m10) / s, (m02 + m20) / s], 1))
    s = torch.sqrt((1 + m11 - m00 - m22).clamp_min(eps)) * 2
    cand.append(torch.stack([(m02 - m20) / s, (m01 + m10) / s, 0.25 * s, (m12 + m21) / s], 1))
    s = torch.sqrt((1 + m22 - m00 - m11).clamp_min(eps)) * 2
    cand.append(torch.stack([(m10 - m01) / s, (m02 + m20) / s, (m12 + m21) / s, 0.25 * s], 1))
    which = torch.where(tr > 0, torch.zeros_like(tr, dtype=torch.long),
                        1 + torch.stack([m00, m11, m22], 1).argmax(1))
    return torch.stack(cand, 1)[torch.arange(m.shape[0], device=R.device), which]
