"""Fixed-grid ODE integrator (reference: LDMAE/transport/integrators.py:77-125; the reference delegates the ODE to
torchdiffeq.odeint(method='euler') -- here the fixed-step solvers are written out, no torchdiffeq)."""
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
