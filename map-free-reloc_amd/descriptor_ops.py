"""SIFT-descriptor correspondence leg on the GPU: rootSIFT + exact 2-NN + Lowe ratio test
(csrc/descriptor_match.hip, include/mfr_hip.h).  Torch only provides memory and the stream.

Reference leg being replaced: everything after `sift.detectAndCompute` in
SIFTMatching.get_correspondences (lib/models/matching/feature_matching.py:86-104) and
SIFT_matcher.match (etc/feature_matching_baselines/matchers.py:160-188): root_sift, the FLANN kd-tree
2-NN (here: the exact search it approximates) and the ratio loop.
"""
import torch

from . import _lib


def _chk(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.MfrLibraryError(f"{name} must be a CUDA(HIP) tensor: the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


def rootsift(desc):
    """[..., 128] f32 raw SIFT descriptors -> (rootSIFT descriptors, squared norms [...])"""
    lib = _lib.load(require_gpu=True)
    desc = _chk(desc, torch.float32, "desc")
    if desc.shape[-1] != 128:
        raise ValueError("SIFT descriptors are 128-d")
    out = torch.empty_like(desc)
    n2 = torch.empty(desc.shape[:-1], dtype=torch.float32, device=desc.device)
    rows = desc.numel() // 128
    _lib.check(lib.mfr_rootsift(_lib.ptr(desc), rows, _lib.ptr(out), _lib.ptr(n2), _lib.stream_ptr()), "mfr_rootsift")
    return out, n2


def ratio_match(des0, des1, norm0, norm1, kp0, kp1, n0, n1, ratio=0.8, max_corr=None):
    """des0 [B,N0,128], des1 [B,N1,128] (rootSIFT), norms, kp0 [B,N0,2], kp1 [B,N1,2], n0/n1 [B] i32.
    Returns dict(pts0 [B,maxN,2], pts1, n_corr [B], nn_idx [B,N0], nn_d2 [B,N0,2])."""
    lib = _lib.load(require_gpu=True)
    des0 = _chk(des0, torch.float32, "des0"); des1 = _chk(des1, torch.float32, "des1")
    norm0 = _chk(norm0, torch.float32, "norm0"); norm1 = _chk(norm1, torch.float32, "norm1")
    kp0 = _chk(kp0, torch.float32, "kp0"); kp1 = _chk(kp1, torch.float32, "kp1")
    n0 = _chk(n0, torch.int32, "n0"); n1 = _chk(n1, torch.int32, "n1")
    B, N0, _ = des0.shape
    N1 = des1.shape[1]
    maxN = int(max_corr or N0)
    dev = des0.device
    nn_idx = torch.full((B, N0), -1, dtype=torch.int32, device=dev)
    nn_d2 = torch.zeros(B, N0, 2, dtype=torch.float32, device=dev)
    pts0 = torch.zeros(B, maxN, 2, dtype=torch.float32, device=dev)
    pts1 = torch.zeros(B, maxN, 2, dtype=torch.float32, device=dev)
    n_corr = torch.zeros(B, dtype=torch.int32, device=dev)
    _lib.check(lib.mfr_desc_ratio_match(_lib.ptr(des0), _lib.ptr(des1), _lib.ptr(norm0), _lib.ptr(norm1), _lib.ptr(kp0),
                                        _lib.ptr(kp1), B, N0, N1, _lib.ptr(n0), _lib.ptr(n1), float(ratio), _lib.ptr(nn_idx),
                                        _lib.ptr(nn_d2), _lib.ptr(pts0), _lib.ptr(pts1), maxN, _lib.ptr(n_corr),
                                        _lib.stream_ptr()), "mfr_desc_ratio_match")
    return dict(pts0=pts0, pts1=pts1, n_corr=n_corr, nn_idx=nn_idx, nn_d2=nn_d2)


class DescriptorRatioMatcher:
    """Batched front door: lists of per-image (keypoints [n,2], raw descriptors [n,128]) for the two
    images of each pair -> correspondences in the solver layout, all on the GPU."""

    def __init__(self, ratio=0.8, device="cuda"):
        self.ratio = float(ratio)
        self.device = torch.device(device)

    def _pack(self, items):
        n = [int(len(k)) for k, _ in items]
        N = max(max(n), 1)
        kp = torch.zeros(len(items), N, 2, dtype=torch.float32)
        de = torch.zeros(len(items), N, 128, dtype=torch.float32)
        for b, (k, d) in enumerate(items):
            if n[b]:
                kp[b, :n[b]] = torch.as_tensor(k, dtype=torch.float32).reshape(-1, 2)
                de[b, :n[b]] = torch.as_tensor(d, dtype=torch.float32).reshape(-1, 128)
        return kp.to(self.device), de.to(self.device), torch.tensor(n, dtype=torch.int32, device=self.device)

    def __call__(self, feats0, feats1):
        kp0, de0, n0 = self._pack(feats0)
        kp1, de1, n1 = self._pack(feats1)
        r0, q0 = rootsift(de0)
        r1, q1 = rootsift(de1)
        return ratio_match(r0, r1, q0, q1, kp0, kp1, n0, n1, self.ratio)
