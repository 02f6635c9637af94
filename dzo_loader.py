"""Imports the package directory ``dzoptimization.jl_amd/`` (its name contains a dot, so a
plain ``import`` cannot reach it) and exposes it as ``dzo``.

    from dzo_loader import dzo
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "dzoptimization.jl_amd")
_NAME = "dzoptimization_jl_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


dzo = load()
