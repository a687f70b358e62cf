"""Import alias: makes the repo directory ``gaussdca.jl_amd/`` importable as the Python
package ``gaussdca.jl_amd`` (a dot cannot appear in a plain package directory name)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gaussdca.jl_amd")
_spec = _ilu.spec_from_file_location("gaussdca.jl_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
jl_amd = _ilu.module_from_spec(_spec)
_sys.modules["gaussdca.jl_amd"] = jl_amd
_spec.loader.exec_module(jl_amd)
