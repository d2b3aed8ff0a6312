"""Import shim: the package directory is named ``abcdez.jl_amd`` (not a valid Python
identifier), so ``import abcdez_amd`` loads it from there under this module name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "abcdez.jl_amd")
_spec = _ilu.spec_from_file_location("abcdez_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["abcdez_amd"] = _mod
_spec.loader.exec_module(_mod)
