"""Fixed-grid ODE / SDE integrators (reference: LDMAE/transport/integrators.py; the reference delegates the
ODE to torchdiffeq.odeint(method='euler') -- here the fixed-step solvers are written out, no torchdiffeq)."""
import torch as th


def shifted_grid(t0, t1, num_steps, timestep_shift):
    """linspace, then t -> s t / (1 + (s-1) t) when s > 0 (integrators.py:92-101)."""
    t = th.linspace(t0, t1, num_steps)
    if timestep_shift > 0:
        t = th.tensor([(timestep_shift * tn) / (1 + (timestep_shift - 1) * tn) for tn in t])
    return t


class ode:
    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol, rtol, timestep_shift=0.0):
        assert t0 < t1, "ODE sampler has to be in forward time"
        self.drift = drift
        self.t = shifted_grid(t0, t1, num_steps, timestep_shift)
        self.atol, self.rtol = atol, rtol
        self.sampler_type = sampler_type.lower()
        if self.sampler_type not in ("euler", "heun", "midpoint"):
            raise NotImplementedError(f"ldmae_amd: fixed-step solvers only (euler / heun / midpoint), got {sampler_type}")

    def sample(self, x, model, **model_kwargs):
        """Returns the stacked trajectory [len(t), ...] like odeint; callers take [-1] (inference.py:287)."""
        t = self.t.to(x.device)

        def f(tk, xk):
            return self.drift(xk, th.ones(xk.size(0), device=xk.device) * tk, model, **model_kwargs)

        xs = [x]
        with th.no_grad():
            for k in range(len(t) - 1):
                dt = t[k + 1] - t[k]
                if self.sampler_type == "euler":
                    x = x + dt * f(t[k], x)
                elif self.sampler_type == "midpoint":
                    x = x + dt * f(t[k] + dt / 2, x + dt / 2 * f(t[k], x))
                else:
                    k1 = f(t[k], x)
                    x = x + dt / 2 * (k1 + f(t[k + 1], x + dt * k1))
                xs.append(x)
        return th.stack(xs)


class sde:
    """Euler-Maruyama / Heun SDE sampler (integrators.py:8-75)."""

    def __init__(self, drift, diffusion, *, t0, t1, num_steps, sampler_type):
        assert t0 < t1, "SDE sampler has to be in forward time"
        self.t = th.linspace(t0, t1, num_steps)
        self.dt = self.t[1] - self.t[0]
        self.drift, self.diffusion, self.sampler_type = drift, diffusion, sampler_type

    def _euler(self, x, t, model, **kw):
        w = th.randn(x.size()).to(x)
        tv = th.ones(x.size(0)).to(x) * t
        mean_x = x + self.drift(x, tv, model, **kw) * self.dt
        return mean_x + th.sqrt(2 * self.diffusion(x, tv)) * w * th.sqrt(self.dt), mean_x

    def _heun(self, x, t, model, **kw):
        w = th.randn(x.size()).to(x)
        tv = th.ones(x.size(0)).to(x) * t
        xhat = x + th.sqrt(2 * self.diffusion(x, tv)) * w * th.sqrt(self.dt)
        k1 = self.drift(xhat, tv, model, **kw)
        k2 = self.drift(xhat + self.dt * k1, tv + self.dt, model, **kw)
        return xhat + 0.5 * self.dt * (k1 + k2), xhat

    def sample(self, init, model, **kw):
        step = {"Euler": self._euler, "Heun": self._heun}.get(self.sampler_type)
        if step is None:
            raise NotImplementedError("Smapler type not implemented.")
        x, out = init, []
        for ti in self.t[:-1]:
            with th.no_grad():
                x, _ = step(x, ti, model, **kw)
                out.append(x)
        return out
