"""VMAE tokenizer mirror (reference package: LDMAE/tokenizer/; only models_mae and util are mirrored, the rest falls through)."""
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
    _sys.modules[__name__] = _dropin.alias(__name__)     # `tokenizer.x` IS `ldmae_amd.tokenizer.x` from here on (ldmae_amd/_dropin.py)
