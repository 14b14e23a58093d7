"""Import alias: `import sdy_amd` loads the package in `spherical-dyffusion_amd/` (a hyphen is not importable)."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spherical-dyffusion_amd")
_spec = importlib.util.spec_from_file_location("sdy_amd", os.path.join(_root, "__init__.py"),
                                               submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sdy_amd"] = _mod
_spec.loader.exec_module(_mod)
