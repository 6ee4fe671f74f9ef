"""Import alias: the package directory is `robust-dynrf_amd/` (not a Python identifier), so
`import rodynrf` loads it through importlib and aliases the module."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("robust-dynrf_amd")
sys.modules[__name__] = _pkg
