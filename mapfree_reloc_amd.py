"""Import alias: the package directory is `map-free-reloc_amd/` (not a valid Python identifier), so
`import mapfree_reloc_amd` resolves to it -- and `mapfree_reloc_amd.<sub>` to the SAME module object as the
package's own relative imports see (`map-free-reloc_amd.<sub>`): one copy of every module, one set of classes,
one loaded library handle."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL, _ALIAS = "map-free-reloc_amd", __name__
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            real = importlib.util.find_spec(_REAL + fullname[len(_ALIAS):])
            return importlib.util.spec_from_loader(fullname, self, origin=real.origin if real else None)
        return None

    def create_module(self, spec):
        mod = importlib.import_module(_REAL + spec.name[len(_ALIAS):])
        self._spec = getattr(mod, "__spec__", None)
        return mod

    def exec_module(self, module):
        if getattr(self, "_spec", None) is not None:       # keep the real module's own spec
            module.__spec__ = self._spec

    def get_code(self, fullname):                          # `python -m mapfree_reloc_amd.<sub>`: run the real module's code
        real = importlib.util.find_spec(_REAL + fullname[len(_ALIAS):])
        return real.loader.get_code(real.name)

    def is_package(self, fullname):
        real = importlib.util.find_spec(_REAL + fullname[len(_ALIAS):])
        return real.submodule_search_locations is not None


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
