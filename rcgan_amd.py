"""Import alias: ``import rcgan_amd`` loads the package kept in ``robust-conditional-gan_amd/``
(a directory name Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robust-conditional-gan_amd")
_spec = importlib.util.spec_from_file_location("rcgan_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rcgan_amd"] = _mod
_spec.loader.exec_module(_mod)
