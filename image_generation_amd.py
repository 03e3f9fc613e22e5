"""Import shim: the product package lives in the directory ``image-generation_amd/``
(a hyphen is not importable), so ``import image_generation_amd`` loads it from there.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-generation_amd")
_spec = importlib.util.spec_from_file_location(
    "image_generation_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["image_generation_amd"] = _mod
_spec.loader.exec_module(_mod)
