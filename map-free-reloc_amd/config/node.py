"""Minimal yacs-compatible CfgNode (yacs is not installed offline).

Behaviour kept from yacs as the reference relies on it (config/default.py:1-115, submission.py:70-71,
config/utils.py:1-11): attribute access, nested nodes, `merge_from_file(yaml)` merging over the
declared defaults, and REJECTION of keys that the schema does not declare ("Non-existent config
key"), plus literal decoding of string values such as 'None' (config/mapfree.yaml:4).
"""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise KeyError(f"config key {full} is not a node")
                self[k]._merge(v, path + [k])
            else:
                self[k] = self._decode(v)

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_file(self, path):
        with open(path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent config key: {k}")
            node[parts[-1]] = self._decode(v)
