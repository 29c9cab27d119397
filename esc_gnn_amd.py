"""Import shim: the package directory is named `esc-gnn_amd/` (not a valid Python identifier),
so `import esc_gnn_amd` loads that directory as the package `esc_gnn_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "esc-gnn_amd")
_spec = importlib.util.spec_from_file_location(
    "esc_gnn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["esc_gnn_amd"] = _mod
_spec.loader.exec_module(_mod)
