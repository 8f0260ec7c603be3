"""Imports the package directory `dynamicsparsearrays.jl_amd/` (whose name is not a valid
Python identifier) as module `dsa_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "dynamicsparsearrays.jl_amd")


def load():
    if "dsa_amd" in sys.modules:
        return sys.modules["dsa_amd"]
    spec = importlib.util.spec_from_file_location(
        "dsa_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dsa_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
