"""build_model(cfg, checkpoint='') -> torch.nn.Module (lib/models/builder.py:8-26).  Only the
'FeatureMatching' family is on the accelerated path; the regression models are out of scope."""
from .matching.model import FeatureMatchingModel


def build_model(cfg, checkpoint=''):
    if cfg.MODEL == 'FeatureMatching':
        return FeatureMatchingModel(cfg)
    raise NotImplementedError(
        f"MODEL={cfg.MODEL!r}: only 'FeatureMatching' is implemented here (Regression models are out of "
        f"scope, SURVEY.md 2 rows 9-10)")
