"""Fused on-GPU relative-pose pipeline for batches of image pairs:

    SuperPoint x2 -> SuperGlue -> depth lift -> PnP-RANSAC (+ refinement)      [config sg_pnp_*]

Replaces, for config/matching/mapfree/sg_pnp_dptkitti.yaml, the reference's two-stage flow
(offline etc/feature_matching_baselines/compute.py:72-86 writing correspondences_SG.npz, then
submission.py:33-58 -> FeatureMatchingModel.forward (lib/models/matching/model.py:29-40) ->
PnPSolver.estimate_pose (pose_solver.py:184-235)) with one device-resident pass: matches never
leave HBM, poses come back as one small tensor per batch.  The npz wire format and the per-pair
plugin API stay available (matching/, wire.py) for drop-in use.
"""
import torch

from . import _lib
from .nets.superpoint import SuperPointHIP
from .nets.superglue import SuperGlueHIP
from .nets import weights as WT
from .solver_ops import PnPBatchSolver


class SuperGluePnPPipeline:
    def __init__(self, device="cuda", sp_state=None, sg_state=None, max_keypoints=1024,
                 pnp_iters=1000, pnp_thr=3.0, pnp_conf=0.9999, seed=0):
        _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.sp = SuperPointHIP(sp_state or WT.superpoint_state_dict(), self.device, max_keypoints=max_keypoints)
        self.sg = SuperGlueHIP(sg_state or WT.superglue_state_dict(), self.device)
        self.pnp = PnPBatchSolver(pnp_iters, pnp_thr, pnp_conf, seed)
        self.K = max_keypoints

    @torch.no_grad()
    def match(self, images):
        """images [2B,1,H,W] -> dict(pts0, pts1 [B,K,2], n_corr [B], ...)  (SuperGlue_matcher.match)"""
        H, W = images.shape[-2:]
        return self.sg(self.sp(images), (H, W), maxN=self.K)

    @torch.no_grad()
    def __call__(self, images, depth0, K0, K1, pair_ids, want_mask=False):
        m = self.match(images)
        out = self.pnp(m["pts0"], m["pts1"], m["n_corr"], depth0, K0, K1, pair_ids, want_mask=want_mask)
        out["n_corr"] = m["n_corr"]
        out["pts0"], out["pts1"] = m["pts0"], m["pts1"]
        return out
