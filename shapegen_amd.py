"""Import alias for the package directory `3d-shape-generation_amd/`.

The directory name required by the repo layout is not a valid Python
identifier, so `import shapegen_amd` loads that directory as a regular package
under this name (relative imports inside it keep working).
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3d-shape-generation_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
