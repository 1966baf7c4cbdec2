import torch as th


def mean_flat(x):
    """Mean over all non-batch dimensions (reference: transport/utils.py:12-16)."""
    return th.mean(x, dim=list(range(1, x.dim())))
