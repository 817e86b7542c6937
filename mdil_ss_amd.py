"""Import alias: ``import mdil_ss_amd`` -> the package living in ``./mdil-ss_amd/`` (a hyphen
is not a valid identifier, so this shim gives the directory an importable name)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mdil-ss_amd")
_spec = _u.spec_from_file_location("mdil_ss_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["mdil_ss_amd"] = _mod
_spec.loader.exec_module(_mod)
