"""Run a reference driver on this tree:  ``python -m ldmae_amd.launch /path/to/reference/LDMAE/train_accum.py --config ...``

The driver's own import lines (``from models.lightningdit import LightningDiT_models``, ``from transport import ...``,
``from tokenizer import models_mae``: LDMAE/train_accum.py:33-37, inference.py:23-28) resolve to the MI355X implementations
(ldmae_amd/_dropin.py); everything this tree does not mirror is served by the reference's own files next to the script.
Works as the target of ``accelerate launch -m ldmae_amd.launch <driver.py> ...`` and ``torchrun -m ldmae_amd.launch <driver.py> ...``.
The runner starts the driver with runpy in THIS process before anything touches the GPU (no exec, no child).
"""
import os
import runpy
import sys

from ldmae_amd import _dropin


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m ldmae_amd.launch <driver.py> [driver args...]")
    script = os.path.abspath(argv[0])
    _dropin.install()
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))          # what `python driver.py` would have put there
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
