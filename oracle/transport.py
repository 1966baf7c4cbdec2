"""Oracle (test infrastructure): flow-matching objective and Euler sampler.

Restates the Linear-path / velocity-prediction branch that the shipped YAML
selects (LDMAE/configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml:52-68).
Citations relative to /root/reference/LDMAE/transport.
"""
from __future__ import annotations

import numpy as np
import torch


def sample_logit_normal(size: int, mu: float = 0.0, sigma: float = 1.0):
    """transport.py:113-123.  scipy ``norm.rvs(loc, scale, size)`` with no
    random_state draws ``numpy.random.standard_normal(size)`` from numpy's
    *global* RandomState and returns ``loc + scale * z`` (scipy 1.15
    rv_continuous.rvs); restated without scipy."""
    z = mu + sigma * np.random.standard_normal(size)
    return torch.tensor(1 / (1 + np.exp(-z)), dtype=torch.float32)


def sample(x1):
    """transport.py:136-166 with use_lognorm=True, train_eps = sample_eps = 0
    (``__init__.py:55-57``): x0 from the torch RNG, t from numpy's RNG, in that order."""
    x0 = torch.randn_like(x1)
    t = sample_logit_normal(x1.shape[0]) * (1 - 0) + 0
    return t.to(x1), x0, x1


def plan(t, x0, x1):
    """path.py:114-136 (ICPlan): xt = t*x1 + (1-t)*x0, ut = x1 - x0."""
    tb = t.view(-1, *([1] * (x1.dim() - 1)))
    return t, tb * x1 + (1 - tb) * x0, 1 * x1 + (-1) * x0


def mean_flat(x):
    """utils.py:12-16."""
    return x.mean(dim=list(range(1, x.dim())))


def training_losses(model_fn, x1, t=None, x0=None):
    """transport.py:169-215, velocity branch.  ``model_fn(xt, t)`` -> prediction."""
    if t is None:
        t, x0, x1 = sample(x1)
    t, xt, ut = plan(t, x0, x1)
    pred = model_fn(xt, t)
    assert pred.shape == xt.shape
    return {"pred": pred, "loss": mean_flat((pred - ut) ** 2)}


def shifted_time_grid(num_steps: int, timestep_shift: float, t0: float = 0.0, t1: float = 1.0):
    """integrators.py:92-101: linspace then t_m = s*t / (1 + (s-1)*t)."""
    t = torch.linspace(t0, t1, num_steps)
    if timestep_shift > 0:
        t = torch.tensor([(timestep_shift * tn) / (1 + (timestep_shift - 1) * tn) for tn in t])
    return t


def euler_ode(drift, x, tgrid):
    """Fixed-grid Euler as torchdiffeq ``odeint(method='euler')`` performs it
    (third-party torchdiffeq, unpinned in requirements.txt; call site
    integrators.py:118-125): x_{k+1} = x_k + (t_{k+1}-t_k) * f(t_k, x_k);
    returns the stacked trajectory including the initial point."""
    xs = [x]
    for k in range(len(tgrid) - 1):
        tk = torch.ones(x.shape[0]) * tgrid[k]
        x = x + (tgrid[k + 1] - tgrid[k]) * drift(x, tk)
        xs.append(x)
    return torch.stack(xs)
