"""Import shim: the product package lives in the directory ``raytracer.glsl_amd/`` (a dotted name
cannot be written in an ``import`` statement), so this module loads it and re-exports it as
``raytracer_glsl_amd``.  ``import raytracer_glsl_amd as rt`` then gives ``rt.scenes``, ``rt.host`` ..."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg_dir = _os.path.join(_here, "raytracer.glsl_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
