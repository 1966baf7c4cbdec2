"""Latent dataset mirror (reference package: LDMAE/datasets/)."""
if __name__.partition(".")[0] != "ldmae_amd":      # imported under the reference's top-level name (PYTHONPATH=<repo>/ldmae_amd)
    import importlib.util as _u
    import os as _os
    import sys as _sys
    if "ldmae_amd" not in _sys.modules:             # make the runtime package importable without touching sys.path
        _r = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
        _sp = _u.spec_from_file_location("ldmae_amd", _os.path.join(_r, "__init__.py"), submodule_search_locations=[_r])
        _m = _u.module_from_spec(_sp)
        _sys.modules["ldmae_amd"] = _m
        _sp.loader.exec_module(_m)
    from ldmae_amd import _dropin
    _sys.modules[__name__] = _dropin.alias(__name__)     # `datasets.x` IS `ldmae_amd.datasets.x` from here on (ldmae_amd/_dropin.py)
else:
    # This package takes the top-level name `datasets` exactly as the reference's LDMAE/datasets/ does -- and so hides the Hugging Face
    # package of that name when it is installed.  accelerate.prepare() (LDMAE/train_accum.py:188) then does
    # `from datasets import IterableDataset` for an isinstance test: give it something to test against instead of an ImportError.
    class IterableDataset:      # never instantiated here
        pass

    class Dataset:
        pass
