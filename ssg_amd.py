"""Import shim: the product package lives in the directory `self-similarity-grouping_amd/`
(a name Python cannot import directly); `import ssg_amd` registers it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "self-similarity-grouping_amd")
_spec = importlib.util.spec_from_file_location("ssg_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ssg_amd"] = _mod
_spec.loader.exec_module(_mod)
