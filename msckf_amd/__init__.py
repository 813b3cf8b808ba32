"""Import shim: the product package lives in `monocular-visual-inertial-msckf_amd/`
(a directory name Python cannot import directly); this module loads it under the
importable name `msckf_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "monocular-visual-inertial-msckf_amd")
_spec = importlib.util.spec_from_file_location(
    "msckf_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["msckf_amd"] = _mod
_spec.loader.exec_module(_mod)
