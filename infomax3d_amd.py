"""Importable alias of the package `3dinfomax_amd/` (a Python identifier cannot start with a digit).

Drop-in use from the reference's train.py: add, after its star-imports (train.py:50-56),

    from infomax3d_amd import *      # PNA, Net3D, NTXent, NTXentMultiplePositives, contrastive_collate, ...

so that the `globals()[...]` plugin lookups (train.py:167-172, 189, 208-209, 589-590) resolve to the MI355X classes.
"""
import importlib as _importlib

_pkg = _importlib.import_module('3dinfomax_amd')
__all__ = list(_pkg.__all__)


def __getattr__(name):
    return getattr(_pkg, name)


def __dir__():
    return __all__
