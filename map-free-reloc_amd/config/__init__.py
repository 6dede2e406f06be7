from .node import CfgNode  # noqa: F401
from .default import get_cfg_defaults, cfg  # noqa: F401
