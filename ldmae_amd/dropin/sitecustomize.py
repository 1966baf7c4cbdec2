"""Start-up hook: with ``PYTHONPATH=<repo>/ldmae_amd/dropin`` every Python process (the ranks that ``accelerate launch`` /
``torchrun`` start from the reference's unchanged ``run_train.sh`` / ``run_inference.sh`` included) resolves the reference's
``models`` / ``transport`` / ``tokenizer`` / ``datasets`` imports to ldmae_amd (ldmae_amd/_dropin.py).  A script's own directory
sits in ``sys.path[0]``, ahead of PYTHONPATH, so shadowing by path order alone cannot work for ``python train_accum.py``;
a meta-path finder can.  Imports nothing heavy (no torch) and chains to the sitecustomize this file shadows."""
import importlib.machinery
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.dirname(_here)
if "ldmae_amd" not in sys.modules:
    _sp = importlib.util.spec_from_file_location("ldmae_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
    _m = importlib.util.module_from_spec(_sp)
    sys.modules["ldmae_amd"] = _m
    _sp.loader.exec_module(_m)
from ldmae_amd import _dropin  # noqa: E402

_dropin.install()
_next = importlib.machinery.PathFinder.find_spec("sitecustomize", [p for p in sys.path if p and os.path.abspath(p) != _here])
if _next is not None and _next.loader is not None:
    _mod = importlib.util.module_from_spec(_next)
    try:
        _next.loader.exec_module(_mod)
    except Exception:      # the shadowed hook is best-effort, as it is for the interpreter itself
        pass
