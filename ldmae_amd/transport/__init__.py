"""Flow-matching transport (stays in Python per the north star; reference: LDMAE/transport/).

Same public surface as the reference package: ``create_transport``, ``Transport``, ``Sampler``,
``ModelType`` / ``PathType`` / ``WeightType``.
"""
from .transport import ModelType, PathType, Sampler, Transport, WeightType


def create_transport(path_type='Linear', prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                     use_cosine_loss=None, use_lognorm=None, partitial_train=None, partial_ratio=1.0, shift_lg=False):
    """Reference: transport/__init__.py:3-72 (same positional order and defaults)."""
    model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE}.get(prediction, ModelType.VELOCITY)
    loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}.get(loss_weight, WeightType.NONE)
    path = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[path_type]
    if path == PathType.VP:
        train_eps, sample_eps = (1e-5 if train_eps is None else train_eps), (1e-3 if train_eps is None else sample_eps)
    elif model_type != ModelType.VELOCITY:
        train_eps, sample_eps = (1e-3 if train_eps is None else train_eps), (1e-3 if train_eps is None else sample_eps)
    else:   # velocity on a linear / GVP path is stable on the whole interval
        train_eps = sample_eps = 0
    return Transport(model_type=model_type, path_type=path, loss_type=loss_type, train_eps=train_eps, sample_eps=sample_eps,
                     use_cosine_loss=use_cosine_loss, use_lognorm=use_lognorm, partitial_train=partitial_train,
                     partial_ratio=partial_ratio, shift_lg=shift_lg)
