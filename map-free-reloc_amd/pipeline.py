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
        sp = self.sp(images)
        m = self.sg(sp, (H, W), maxN=self.K)
        m["n_kpts"] = sp["n"]
        return m

    @torch.no_grad()
    def __call__(self, images, depth0, K0, K1, pair_ids, want_mask=False):
        m = self.match(images)
        out = self.pnp(m["pts0"], m["pts1"], m["n_corr"], depth0, K0, K1, pair_ids, want_mask=want_mask)
        out["n_corr"] = m["n_corr"]
        out["pts0"], out["pts1"], out["n_kpts"] = m["pts0"], m["pts1"], m["n_kpts"]
        return out


class LoFTREmatPipeline:
    """LoFTR coarse-to-fine matching -> E-matrix RANSAC -> metric scale from depth
    (config/matching/mapfree/loftr_emat_dptkitti.yaml: PIX_THRESHOLD 2.0, SCALE_THRESHOLD 0.1,
    CONFIDENCE 0.9999), one device-resident pass over a batch of pairs.  Replaces
    LoFTR_matcher.match (matchers.py:24-59) + EssentialMatrixMetricSolver (pose_solver.py:115-172)."""

    def __init__(self, device="cuda", loftr_state=None, pix_thr=2.0, scale_thr=0.1, conf=0.9999, seed=0, pad_to=8):
        from .nets.loftr import LoFTRHIP
        from .solver_ops import EssentialBatchSolver, ScaleFromDepthBatch
        _lib.load(require_gpu=True)
        self.device = torch.device(device)
        self.loftr = LoFTRHIP(loftr_state or WT.loftr_state_dict(), self.device)
        self.emat = EssentialBatchSolver(pix_thr, conf, seed)
        self.scale = ScaleFromDepthBatch(scale_thr)
        self.pad_to = pad_to

    @torch.no_grad()
    def match(self, images):
        H, W = images.shape[-2:]
        # LoFTR needs multiples of 8; the reference right-pads W 540 -> 544 (matchers.py:41-46, quirk Q3)
        ph, pw = (-H) % self.pad_to, (-W) % self.pad_to
        if ph or pw:
            images = torch.nn.functional.pad(images, (0, pw, 0, ph))
        return self.loftr(images)

    @torch.no_grad()
    def __call__(self, images, depth0, depth1, K0, K1, pair_ids):
        m = self.match(images)
        e = self.emat(m["pts0"], m["pts1"], m["n_corr"], K0, K1, pair_ids)
        s = self.scale(m["pts0"], m["pts1"], e["mask"], m["n_corr"], depth0, depth1, K0, K1, e["R"], e["t"], e["status"])
        return dict(R=torch.where((s["status"] == 0)[:, None, None], e["R"], torch.full_like(e["R"], float("nan"))),
                    t=s["t_metric"], n_inliers=s["n_inliers"], status=s["status"], n_corr=m["n_corr"],
                    emat_inliers=e["n_inliers"], pts0=m["pts0"], pts1=m["pts1"], emat_mask=e["mask"])
