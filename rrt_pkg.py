"""Imports the host package that lives in ``rust-ray-tracer_amd/`` (a directory name Python cannot import
directly because of the hyphen) under the module name ``rust_ray_tracer_amd``."""
import importlib.util
import os
import sys

_NAME = "rust_ray_tracer_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "rust-ray-tracer_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    path = os.path.join(PKG_DIR, "__init__.py")
    spec = importlib.util.spec_from_file_location(_NAME, path, submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
