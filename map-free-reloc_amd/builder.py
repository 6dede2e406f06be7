"""build_model(cfg, checkpoint='') -> torch.nn.Module (lib/models/builder.py:8-27): the feature-matching family
(SURVEY.md 8 rows a-*) and the regression family (row f-4; `checkpoint` = Lightning checkpoint of the reference)."""
from .matching.model import FeatureMatchingModel


def build_model(cfg, checkpoint=''):
    from . import options
    options.apply_cfg(cfg)                         # declared kernel-selection options under cfg.HIP (options.py)
    if cfg.MODEL == 'FeatureMatching':
        return FeatureMatchingModel(cfg)
    if cfg.MODEL in ('Regression', 'RegressionMultiFrame'):
        from .regression.model import build_regression_model
        return build_regression_model(cfg, checkpoint)
    raise NotImplementedError(f"MODEL={cfg.MODEL!r}")
