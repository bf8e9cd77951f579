"""Loader shim: makes the directory ``cortex.jl_amd/`` importable as the package ``cortex.jl_amd``
(a dotted directory name cannot be imported directly)."""
import importlib.util as _u
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cortex.jl_amd")
if "cortex.jl_amd" not in _sys.modules:
    _spec = _u.spec_from_file_location("cortex.jl_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                       submodule_search_locations=[_pkg_dir])
    jl_amd = _u.module_from_spec(_spec)
    _sys.modules["cortex.jl_amd"] = jl_amd
    _spec.loader.exec_module(jl_amd)
else:
    jl_amd = _sys.modules["cortex.jl_amd"]
