"""Import shim: the package directory is named after the reference repository
(`unige-tasi-path-planners_amd`, not a valid Python identifier), so it is
loaded here under the module name `ufm_amd_pkg` and re-exported."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unige-tasi-path-planners_amd")
_NAME = "ufm_amd_pkg"
if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
_pkg = sys.modules[_NAME]
globals().update({k: getattr(_pkg, k) for k in dir(_pkg) if not k.startswith("__")})
PKG_DIR = _PKG_DIR
