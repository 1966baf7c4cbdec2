"""ldmae_amd -- MI355X-native (gfx950) kernels and host-side module mirror for the LDMAE hot path:
the LightningDiT flow-matching train step and the VMAE masked-token encoder.

Layout: ``csrc/`` hand-written HIP kernels + C ABI (``include/ldmae_hip.h``); ``_lib`` ctypes binding;
``ops`` tensor-level wrappers; ``models/``, ``tokenizer/``, ``transport/`` mirror the reference's module API
(put this directory on PYTHONPATH to shadow the reference's packages -- INTEGRATION.md).
"""
__version__ = "0.1.0"
