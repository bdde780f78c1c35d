"""Import shim: the package directory is named `katana.jl_amd/` (a dot is not importable),
so `import katana_jl_amd` loads it from that directory under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "katana.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "katana_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["katana_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
