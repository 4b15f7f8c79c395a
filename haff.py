"""Import alias: `import haff` == the package in ./2handedafforder_amd (whose name is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("2handedafforder_amd")
sys.modules[__name__] = _pkg
