"""Drop-in name resolution: makes the reference drivers' own import lines

    from models.lightningdit import LightningDiT_models           # LDMAE/train_accum.py:33, inference.py:26
    from transport import create_transport, Sampler               # LDMAE/train_accum.py:34, inference.py:27
    from tokenizer import models_mae                              # LDMAE/inference.py:23, extract_features.py:13
    from datasets.img_latent_dataset import ImgLatentDataset      # LDMAE/train_accum.py:36
    from tokenizer.util.misc import DiagonalGaussianDistribution  # LDMAE/train_accum.py:37

resolve to this tree.  ``install()`` puts ONE finder at the head of ``sys.meta_path``: a module ``<root>.<x>`` that this tree
mirrors is served as an ALIAS of ``ldmae_amd.<root>.<x>`` (the same module object under both names: one set of classes, one
copy of the kernel-library state); a submodule this tree does not mirror (``tokenizer.vavae``, ``tokenizer.sdvae``, ...) falls
through to the same-named package directories found on ``sys.path`` -- the reference's own files keep serving them.

Three ways in:
  * ``python -m ldmae_amd.launch <reference driver.py> args...``   (explicit; works under ``accelerate launch -m`` / ``torchrun -m``)
  * ``PYTHONPATH=<repo>/ldmae_amd/dropin``: Python imports ``dropin/sitecustomize.py`` at start-up, which calls ``install()`` --
    ``run_train.sh`` / ``run_inference.sh`` run unchanged, the script directory in ``sys.path[0]`` notwithstanding
  * ``PYTHONPATH=<repo>/ldmae_amd`` for code that is not run as a script from the reference directory (``python -c``, notebooks):
    the mirrored packages' ``__init__`` call ``install()`` themselves.

Nothing here touches the GPU or imports torch.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

MIRRORED = ("models", "transport", "tokenizer", "datasets")
_PKG = "ldmae_amd"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        mod = importlib.import_module(self.target)       # the runtime module itself: `models.x is ldmae_amd.models.x`
        self._spec = getattr(mod, "__spec__", None)      # importlib is about to overwrite it with the ALIAS spec ...
        return mod

    def exec_module(self, module):
        # ... put the module's own spec back: with __spec__.parent = 'models' and __package__ = 'ldmae_amd.models' a later relative import
        # inside the aliased module warns (ImportWarning; an error under -W error) or, on interpreters that resolve relative imports through
        # __spec__.parent, looks in the wrong package
        if getattr(self, "_spec", None) is not None:
            module.__spec__ = self._spec


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.partition(".")[0]
        if root not in MIRRORED:
            return None
        tgt = f"{_PKG}.{fullname}"
        try:
            ours = importlib.util.find_spec(tgt)
        except (ImportError, ValueError):
            ours = None
        if ours is not None:
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(tgt), origin=ours.origin,
                                                  is_package=ours.submodule_search_locations is not None)
        parent = fullname.rpartition(".")[0]
        if not parent:
            return None
        # not mirrored here: the reference's own file (any same-named package directory on sys.path) serves it
        here = os.path.dirname(os.path.abspath(__file__))
        dirs = []
        for p in sys.path:
            d = os.path.join(p or ".", *parent.split("."))
            if os.path.isdir(d) and not os.path.abspath(d).startswith(here + os.sep):
                dirs.append(d)
        return importlib.machinery.PathFinder.find_spec(fullname, dirs) if dirs else None


def install() -> None:
    """Idempotent.  Modules already imported under a mirrored top-level name from somewhere else are left alone."""
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, _Finder())


def alias(name: str):
    """The runtime module behind a mirrored top-level name (used by the packages' own ``__init__`` in PYTHONPATH-shadow mode)."""
    install()
    return importlib.import_module(f"{_PKG}.{name}")
