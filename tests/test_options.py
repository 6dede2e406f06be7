"""Kernel-selection options are DECLARED (mapfree_reloc_amd/options.py + cfg.HIP.*), never read from the environment."""
import os
import re

import pytest

import mapfree_reloc_amd  # noqa: F401
from mapfree_reloc_amd import options
from mapfree_reloc_amd.config import get_cfg_defaults

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_options_are_declared_validated_and_in_the_config_schema():
    cfg = get_cfg_defaults()
    for k in options.names():
        assert k in cfg.HIP and cfg.HIP[k] == options.default(k)
    with pytest.raises(KeyError):
        options.set("NOT_AN_OPTION", 1)
    with pytest.raises(ValueError):
        options.set("CONV_KERNEL", "fastest")
    try:
        cfg.HIP.CONV_KERNEL = "exact"; cfg.HIP.RPR_WGRAD_SPLITS = 8
        options.apply_cfg(cfg)
        assert options.get("CONV_KERNEL") == "exact" and options.get("RPR_WGRAD_SPLITS") == 8
        from mapfree_reloc_amd.nets.conv import prefer_split
        assert prefer_split(720, 540) is False
        options.set("CONV_KERNEL", "split")
        assert prefer_split(90, 67) is True
    finally:
        options.reset()
    assert options.get("CONV_KERNEL") == "auto"


def test_directly_set_option_survives_a_configuration_that_carries_the_default():
    """bench.py --hip-opt / a test sets an option, then the trainer or predict_fused applies its configuration (which declares EVERY option
    with its default): the direct setting stays; a configuration that really asks for another value wins; reset() forgets both"""
    cfg = get_cfg_defaults()
    try:
        options.set("RPR_CONV", "miopen")
        options.apply_cfg(cfg)                              # cfg.HIP.RPR_CONV == 'hip' (the default)
        assert options.get("RPR_CONV") == "miopen"
        cfg.HIP.RPR_CONV_ORDER = "tap_outer"                # a value the configuration sets itself
        options.apply_cfg(cfg)
        assert options.get("RPR_CONV_ORDER") == "tap_outer" and options.get("RPR_CONV") == "miopen"
        cfg.HIP.RPR_CONV_ORDER = "tap_inner"                # ... and takes back: not a direct setting, so the configuration decides
        options.apply_cfg(cfg)
        assert options.get("RPR_CONV_ORDER") == "tap_inner"
    finally:
        options.reset()
    assert options.get("RPR_CONV") == "hip"


def test_product_reads_no_mfr_environment_variable():
    """the only environment the package looks at is the launcher's rendezvous (RANK / WORLD_SIZE / LOCAL_* / MASTER_*)"""
    allowed = {"RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"}
    pkg = os.path.join(ROOT, "map-free-reloc_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if not f.endswith((".py", ".hip", ".h")):
                continue
            src = open(os.path.join(dp, f)).read()
            assert "getenv" not in src or f.endswith(".py"), f
            for m in re.finditer(r"environ(?:\.get|\.setdefault)?\s*[\[(]\s*['\"]([A-Z_0-9]+)['\"]", src):
                assert m.group(1) in allowed, (f, m.group(1))
            for m in re.finditer(r"['\"]([A-Z_0-9]+)['\"]\s+in\s+os\.environ", src):
                assert m.group(1) in allowed, (f, m.group(1))
