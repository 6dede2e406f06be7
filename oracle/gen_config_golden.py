"""Generate tests/golden/ref_config_defaults.json by EXECUTING THE REFERENCE'S OWN config/default.py (build container only), with
yacs (not installed offline) replaced by this package's yacs-compatible CfgNode: every legal key of the reference's schema and its
default value, flattened.  tests/test_data_golden.py checks that this package's schema holds every one of them with the same default."""
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def flat(node, prefix=""):
    from mapfree_reloc_amd.config.node import CfgNode
    out = {}
    for k, v in node.items():
        if isinstance(v, CfgNode):
            out.update(flat(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


def main():
    from mapfree_reloc_amd.config.node import CfgNode
    y, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
    yc.CfgNode = CfgNode
    y.config = yc
    sys.modules["yacs"], sys.modules["yacs.config"] = y, yc
    spec = importlib.util.spec_from_file_location("ref_default", "/root/reference/config/default.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ref = flat(m.cfg)
    json.dump(ref, open(os.path.join(ROOT, "tests", "golden", "ref_config_defaults.json"), "w"), indent=1, sort_keys=True)
    print(len(ref), "keys")


if __name__ == "__main__":
    main()
