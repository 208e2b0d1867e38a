"""Import shim: the package directory is `animal-vision_amd/` (not a valid Python identifier), so
`import animal_vision_amd` lands here and is redirected to that directory as a regular package."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "animal-vision_amd")
_spec = importlib.util.spec_from_file_location(
    "animal_vision_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["animal_vision_amd"] = _mod
_spec.loader.exec_module(_mod)
