"""Import alias: `import haff` == the package in ./2handedafforder_amd (whose name is not a Python identifier).
`haff.x` and `2handedafforder_amd.x` are the SAME module object: a meta-path finder maps every `haff.*` import onto the
real module (without it `from haff.x import y` would execute a second copy of x — and of everything x imports
relatively — so patching or isinstance checks through one name would miss the other)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "2handedafforder_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith("haff."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        real = importlib.import_module(_REAL + spec.name[len("haff"):])
        self._spec = real.__spec__          # module_from_spec overwrites __spec__: put the real one back in exec_module
        return real

    def exec_module(self, module):
        module.__spec__ = self._spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
