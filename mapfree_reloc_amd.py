"""Import alias: the package directory is `map-free-reloc_amd/` (not a valid Python identifier),
so `import mapfree_reloc_amd` resolves to it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("map-free-reloc_amd")
sys.modules[__name__] = _pkg
