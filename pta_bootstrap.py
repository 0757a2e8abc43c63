"""Import helper: the product package lives in the directory ``pose-transfer_amd/``
(name fixed by the build contract).  A hyphen is not a legal Python identifier, so
this module registers that directory under the importable alias ``pose_transfer_amd``.

    import pta_bootstrap; pta = pta_bootstrap.load()
    from pose_transfer_amd.models.networks import Deformable_Generator
"""
import importlib.util
import os
import sys

_ALIAS = "pose_transfer_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "pose-transfer_amd")


def load():
    mod = sys.modules.get(_ALIAS)
    if mod is not None:
        return mod
    spec = importlib.util.spec_from_file_location(
        _ALIAS, os.path.join(PKG_DIR, "__init__.py"),
        submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
