"""Import shim: the package directory is named `climaseaice.jl_amd` (with a dot), which Python's
import statement cannot spell.  `import climaseaice_jl_amd as csi` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "climaseaice.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
