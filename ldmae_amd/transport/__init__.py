"""Flow-matching transport (stays in Python per the north star; reference: LDMAE/transport/).

Same public surface as the reference package: ``create_transport``, ``Transport``, ``Sampler``,
``ModelType`` / ``PathType`` / ``WeightType``.
"""
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
    _sys.modules[__name__] = _dropin.alias(__name__)     # `transport.x` IS `ldmae_amd.transport.x` from here on (ldmae_amd/_dropin.py)
else:
    from .transport import ModelType, PathType, Sampler, Transport, WeightType


    def create_transport(path_type='Linear', prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                         use_cosine_loss=None, use_lognorm=None, partitial_train=None, partial_ratio=1.0, shift_lg=False):
        """Reference: transport/__init__.py:3-72 (same positional order and defaults).  The shipped configuration -- linear path, velocity
        prediction, unweighted loss -- is the hot path; the other plans / parametrisations of the reference are out of scope."""
        if path_type != "Linear" or prediction != "velocity" or loss_weight not in (None, "None", "none"):
            raise NotImplementedError(f"ldmae_amd transport: only path_type='Linear', prediction='velocity', loss_weight=None are on the hot path "
                                      f"(got {path_type!r}, {prediction!r}, {loss_weight!r}); SURVEY.md 2.1 #7")
        # velocity on the linear path is stable on the whole interval (transport/__init__.py:60-62)
        return Transport(model_type=ModelType.VELOCITY, path_type=PathType.LINEAR, loss_type=WeightType.NONE, train_eps=0, sample_eps=0,
                         use_cosine_loss=use_cosine_loss, use_lognorm=use_lognorm, partitial_train=partitial_train,
                         partial_ratio=partial_ratio, shift_lg=shift_lg)
