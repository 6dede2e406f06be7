"""Relative-pose regression ("3PRegress"): Siamese encoder -> correlation-volume warping
(csrc/corr_warp.hip, fused forward AND backward) -> residual head -> (R, t); bf16 training, one process per GPU.
Mirrors lib/models/regression/ of the reference: same class names, config keys and state-dict keys."""
from .model import RegressionModel, RegressionMultiFrameModel  # noqa: F401
