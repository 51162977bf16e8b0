"""Import alias: `import set_amd` loads the package that lives in the directory
`speech-editing-toolkit_amd/` (its name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "speech-editing-toolkit_amd")
_spec = importlib.util.spec_from_file_location(
    "set_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["set_amd"] = _mod
_spec.loader.exec_module(_mod)
