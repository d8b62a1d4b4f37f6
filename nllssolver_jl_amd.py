"""Import alias for the package directory ``nllssolver.jl_amd/``.

The directory name (fixed by the project layout) contains a dot, which Python's import system
reads as a sub-module separator.  ``import nllssolver_jl_amd`` executes this shim, which loads the
real package from ``nllssolver.jl_amd/__init__.py`` and installs it under this module name.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "nllssolver.jl_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
