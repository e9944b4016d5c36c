"""Import shim: the package directory is named ``rtl-power-fftw_amd`` (not a
valid Python identifier), so this module loads it under the importable name
``rtl_power_fftw_amd``."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rtl-power-fftw_amd")
_spec = importlib.util.spec_from_file_location(
    "rtl_power_fftw_amd",
    os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rtl_power_fftw_amd"] = _mod
_spec.loader.exec_module(_mod)
