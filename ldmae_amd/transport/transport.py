"""Transport objective + samplers (reference: LDMAE/transport/transport.py)."""
import enum

import numpy as np
import torch as th

from . import path
from .integrators import ode
from .utils import mean_flat


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


class Transport:
    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps, use_cosine_loss=False, use_lognorm=False,
                 partitial_train=None, partial_ratio=1.0, shift_lg=False):
        self.loss_type, self.model_type = loss_type, model_type
        if path_type != PathType.LINEAR or model_type != ModelType.VELOCITY:
            raise NotImplementedError("ldmae_amd Transport: linear path + velocity prediction only (SURVEY.md 2.1 #7)")
        self.path_sampler = path.ICPlan()
        self.train_eps, self.sample_eps = train_eps, sample_eps
        self.use_cosine_loss, self.use_lognorm = use_cosine_loss, use_lognorm
        self.partitial_train, self.partial_ratio, self.shift_lg = partitial_train, partial_ratio, shift_lg

    def prior_logp(self, z):
        n = int(np.prod(z.shape[1:]))
        return -n / 2.0 * np.log(2 * np.pi) - th.sum(z.flatten(1) ** 2, dim=1) / 2.0

    def check_interval(self, train_eps, sample_eps, *, diffusion_form="SBDM", sde=False, reverse=False, eval=False,
                       last_step_size=0.0):
        """transport.py:84-111 for the velocity model on the linear path: the whole unit interval."""
        if sde:
            raise NotImplementedError("ldmae_amd Transport: SDE sampling is out of scope (SURVEY.md 2.1 #7)")
        return (1, 0) if reverse else (0, 1)

    def sample_logit_normal(self, mu, sigma, size=1):
        """transport.py:113-123.  The reference calls scipy.stats.norm.rvs with no random_state, i.e.
        numpy's GLOBAL RandomState; `mu + sigma * standard_normal(size)` consumes that stream identically."""
        z = mu + sigma * np.random.standard_normal(size)
        return th.tensor(1 / (1 + np.exp(-z)), dtype=th.float32)

    def sample_in_range(self, mu, sigma, target_size, range_min=0, range_max=0.5):
        out = []
        while len(out) < target_size:
            s = self.sample_logit_normal(mu, sigma, size=target_size)
            out.extend(s[(s >= range_min) & (s <= range_max)])
        return th.tensor(out[:target_size])

    def sample(self, x1, sp_timesteps=None, shifted_mu=0):
        """transport.py:136-166: x0 on x1's device RNG, t on the host (numpy for logit-normal)."""
        x0 = th.randn_like(x1)
        t0, t1 = self.check_interval(self.train_eps, self.sample_eps)
        B = x1.shape[0]
        if not self.use_lognorm:
            if self.partitial_train is not None and th.rand(1) < self.partial_ratio:
                t = th.rand((B,)) * (self.partitial_train[1] - self.partitial_train[0]) + self.partitial_train[0]
            else:
                t = th.rand((B,)) * (t1 - t0) + t0
        elif not self.shift_lg:
            if self.partitial_train is not None and th.rand(1) < self.partial_ratio:
                t = self.sample_in_range(0, 1, B, range_min=self.partitial_train[0], range_max=self.partitial_train[1])
            else:
                t = self.sample_logit_normal(0, 1, size=B) * (t1 - t0) + t0
        else:
            assert self.partitial_train is None, "Shifted lognormal distribution is not compatible with partial training"
            t = self.sample_logit_normal(shifted_mu, 1, size=B) * (t1 - t0) + t0
        if sp_timesteps is not None:
            t = th.rand((B,)) * (sp_timesteps[1] - sp_timesteps[0]) + sp_timesteps[0]
        # host-drawn t -> device through pinned memory, asynchronously: a pageable .to(device) makes torch synchronise the stream, i.e. the
        # host waits for the whole previous step and the GPU then idles (~1 ms per step) while the next step's first kernels are launched
        if x1.is_cuda and not t.is_cuda:
            t = t.to(x1.dtype).pin_memory().to(x1.device, non_blocking=True)
        else:
            t = t.to(x1)
        return t, x0, x1

    def training_losses(self, model, x1, model_kwargs=None, sp_timesteps=None, shifted_mu=0):
        """transport.py:169-215."""
        model_kwargs = model_kwargs or {}
        t, x0, x1 = self.sample(x1, sp_timesteps, shifted_mu)
        t, xt, ut = self.path_sampler.plan(t, x0, x1)
        out = model(xt, t, **model_kwargs)
        assert out.size() == xt.size()
        terms = {'pred': out}
        terms['loss'] = mean_flat((out - ut) ** 2)
        if self.use_cosine_loss:
            terms['cos_loss'] = mean_flat(1 - th.nn.functional.cosine_similarity(out, ut, dim=1))
        return terms

    def get_drift(self):
        """transport.py:217-245, velocity model: the drift of the probability-flow ODE is the model output itself."""
        def body_fn(x, t, model, **kw):
            out = model(x, t, **kw)
            assert out.shape == x.shape, "Output shape from ODE solver must match input shape"
            return out
        return body_fn


class Sampler:
    """transport.py:270-443, ODE sampler (the SDE sampler and likelihood evaluation are out of scope)."""

    def __init__(self, transport):
        self.transport = transport
        self.drift = transport.get_drift()

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False, timestep_shift=0.0):
        drift = (lambda x, t, model, **kw: self.drift(x, th.ones_like(t) * (1 - t), model, **kw)) if reverse else self.drift
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False, eval=True,
                                               reverse=reverse, last_step_size=0.0)
        return ode(drift=drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol, rtol=rtol,
                   timestep_shift=timestep_shift).sample

    def sample_sde(self, **_):
        raise NotImplementedError("ldmae_amd Sampler: SDE sampling is out of scope (SURVEY.md 2.1 #7); run_inference.sh uses sample_ode")
