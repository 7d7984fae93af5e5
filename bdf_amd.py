"""Import shim: the package directory is named `bayesiandatafusion.jl_amd` (a dot cannot appear in a Python
module name), so `import bdf_amd` loads that directory as the package `bdf_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bayesiandatafusion.jl_amd")
_spec = importlib.util.spec_from_file_location("bdf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bdf_amd"] = _mod
_spec.loader.exec_module(_mod)
