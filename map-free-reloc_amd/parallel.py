"""Multi-GPU layer: one process per GPU, image pairs sharded embarrassingly.

The reference is a single process (submission.py:33-58 loops pairs serially, batch 1;
train.py:53 devices=1) and has no collective.  Here each rank runs the whole hot path on its
own contiguous shard of the (scene, frame)-ordered pair list; the ONLY collective is one
all_gather (RCCL over xGMI on GPUs, gloo in the CPU tests) of fixed-width per-pair pose records
at the end of a run: [pair_id, qw, qx, qy, qz, tx, ty, tz, inliers, status] as 10 x f64 = 80 B per
pair (~1.2 MB for the whole Map-free test split) -- latency-bound, issued once, never per pair.
"""
import torch
import torch.distributed as dist

REC_W = 10


def shard_range(n_items, world, rank):
    """contiguous, balanced [lo, hi) of rank's items"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_scenes(pairs_per_scene, world):
    """contiguous scene blocks balanced by pair count (keeps each scene's npz / depth maps and its
    pose_{scene}.txt on one rank).  Returns list of (scene_lo, scene_hi) per rank."""
    total = sum(pairs_per_scene)
    out, lo, acc = [], 0, 0
    n = len(pairs_per_scene)
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        while hi < n and (acc + pairs_per_scene[hi] <= target + 1e-9 or hi == lo) and (n - hi) > (world - r - 1):
            acc += pairs_per_scene[hi]
            hi += 1
        if r == world - 1:
            hi = n
        out.append((lo, hi))
        lo = hi
    return out


def rotmat_to_quat(R):
    """[...,3,3] -> [...,4] (w,x,y,z), w >= 0: same convention as transforms3d.quaternions.mat2quat
    used by submission.py:53 (restated: transforms3d is not available offline)."""
    m00, m01, m02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    m10, m11, m12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    m20, m21, m22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    tr = m00 + m11 + m22
    qw = torch.sqrt(torch.clamp(1 + tr, min=0)) / 2
    qx = torch.sqrt(torch.clamp(1 + m00 - m11 - m22, min=0)) / 2
    qy = torch.sqrt(torch.clamp(1 - m00 + m11 - m22, min=0)) / 2
    qz = torch.sqrt(torch.clamp(1 - m00 - m11 + m22, min=0)) / 2
    # pick the largest component as pivot for numerical stability
    q = torch.stack([qw, qx, qy, qz], -1)
    piv = q.argmax(-1)
    w0 = torch.stack([qw, (m21 - m12) / (4 * qw), (m02 - m20) / (4 * qw), (m10 - m01) / (4 * qw)], -1)
    x0 = torch.stack([(m21 - m12) / (4 * qx), qx, (m01 + m10) / (4 * qx), (m02 + m20) / (4 * qx)], -1)
    y0 = torch.stack([(m02 - m20) / (4 * qy), (m01 + m10) / (4 * qy), qy, (m12 + m21) / (4 * qy)], -1)
    z0 = torch.stack([(m10 - m01) / (4 * qz), (m02 + m20) / (4 * qz), (m12 + m21) / (4 * qz), qz], -1)
    cand = torch.stack([w0, x0, y0, z0], -2)
    out = torch.gather(cand, -2, piv[..., None, None].expand(*piv.shape, 1, 4)).squeeze(-2)
    sign = torch.where(out[..., :1] < 0, -1.0, 1.0)
    return out * sign


def pose_records(pair_ids, out):
    """device tensors -> [n, 10] f64 records.  R and t pass through float32 first, exactly like the model plugin's
    outputs do (lib/models/matching/model.py:38-39 `.float()`) before submission.py:53 turns R into a quaternion --
    so a record formats to the same text line as the per-pair path."""
    R, t = out["R"].to(torch.float32).to(torch.float64), out["t"].to(torch.float32).to(torch.float64)
    q = rotmat_to_quat(R)
    q = torch.where(torch.isnan(R).any(-1).any(-1, keepdim=True), torch.full_like(q, float("nan")), q)
    return torch.cat([pair_ids.to(torch.float64)[:, None], q, t, out["n_inliers"].to(torch.float64)[:, None],
                      out["status"].to(torch.float64)[:, None]], 1).contiguous()


def gather_pose_records(pair_ids, out, world=None):
    """all ranks -> every rank gets the concatenated [sum n, 10] records (padding rows removed)"""
    return gather_records(pose_records(pair_ids, out), world)


def gather_records(rec, world=None):
    """the one collective of the path: ragged [n_r, 10] record blocks of all ranks -> [sum n_r, 10] on every rank"""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if not dist.is_initialized():
        return rec
    n = torch.tensor([rec.shape[0]], device=rec.device, dtype=torch.int64)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    nmax = int(max(int(x) for x in ns))
    pad = torch.full((nmax, REC_W), float("nan"), dtype=torch.float64, device=rec.device)
    pad[: rec.shape[0]] = rec
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: int(k)] for b, k in zip(bufs, ns)], 0)
