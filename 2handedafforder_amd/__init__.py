"""2handedafforder_amd — MI355X-native hot path of pearl-robot-lab/2HandedAfforder (2Haff/ affordance prediction).

The directory name starts with a digit, so import it as `import haff` (root-level alias module) or
`importlib.import_module("2handedafforder_amd")`.
"""
from .lib import EXPORTED_SYMBOLS, HaffLibraryError, LIB_PATH, build_library, load_library  # noqa: F401

__all__ = ["EXPORTED_SYMBOLS", "HaffLibraryError", "LIB_PATH", "build_library", "load_library"]
