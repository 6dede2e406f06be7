"""FeatureMatchingModel -- the model plugin of the matching path (contract: lib/models/matching/model.py:7-40):

    build_model(cfg)(data) -> (R [1,3,3] f32, t [1,1,3] f32), and data['inliers'] = the solver's confidence.

Two registries replace the reference's dispatch chains; `cfg.FEATURE_MATCHING` / `cfg.POSE_SOLVER` name the
entries, so a new correspondence source or solver is one `register_*` call.  Batch size is 1 on this surface
(the reference's contract); submission.predict_fused / pipeline.FusedPosePipeline are the batched twins.
"""
import numpy as np
import torch

from . import feature_matching as _fm
from . import pose_solver as _ps

FEATURE_MATCHERS = {
    'SIFT': _fm.SIFTMatching,
    'Precomputed': _fm.PrecomputedMatching,
    'SuperGlue': _fm.SuperGlueMatching,      # online, GPU (new)
    'LoFTR': _fm.LoFTRMatching,              # online, GPU (new)
}
POSE_SOLVERS = {
    'EssentialMatrix': _ps.EssentialMatrixSolver,
    'EssentialMatrixMetric': _ps.EssentialMatrixMetricSolver,
    'Procrustes': _ps.ProcrustesSolver,
    'PNP': _ps.PnPSolver,
}


def register_feature_matcher(name, cls):
    FEATURE_MATCHERS[name] = cls


def register_pose_solver(name, cls):
    POSE_SOLVERS[name] = cls


def _lookup(table, key, what):
    if key not in table:
        raise NotImplementedError(f'Invalid {what}: {key!r} (known: {sorted(table)})')
    return table[key]


class FeatureMatchingModel(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.feature_matching = _lookup(FEATURE_MATCHERS, cfg.FEATURE_MATCHING, 'feature matching')(cfg)
        self.pose_solver = _lookup(POSE_SOLVERS, cfg.POSE_SOLVER, 'pose solver')(cfg)

    def forward(self, data):
        if data['depth0'].shape[0] != 1:
            raise AssertionError('Baseline models require batch size of 1')
        kpts0, kpts1 = self.feature_matching.get_correspondences(data)
        R, t, confidence = self.pose_solver.estimate_pose(kpts0, kpts1, data)
        data['inliers'] = confidence
        as_f32 = lambda a, shape: torch.from_numpy(np.array(a, dtype=np.float32, copy=True).reshape(shape))
        return as_f32(R, (1, 3, 3)), as_f32(t, (1, 1, 3))
