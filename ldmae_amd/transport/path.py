"""Coupling plans x_t = alpha_t x1 + sigma_t x0 (reference: LDMAE/transport/path.py)."""
import math

import torch as th


def expand_t_like_x(t, x):
    """[B] -> [B, 1, 1, ...] (path.py:5-13)."""
    return t.view(t.size(0), *([1] * (x.dim() - 1)))


class ICPlan:
    """Linear path: alpha = t, sigma = 1 - t (path.py:18-136)."""

    def __init__(self, sigma=0.0):
        self.sigma = sigma

    def compute_alpha_t(self, t):
        return t, 1

    def compute_sigma_t(self, t):
        return 1 - t, -1

    def compute_d_alpha_alpha_ratio_t(self, t):
        return 1 / t

    def compute_drift(self, x, t):
        t = expand_t_like_x(t, x)
        ratio = self.compute_d_alpha_alpha_ratio_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        return -(ratio * x), ratio * (sigma_t ** 2) - sigma_t * d_sigma_t

    def compute_diffusion(self, x, t, form="constant", norm=1.0):
        t = expand_t_like_x(t, x)
        forms = {
            "constant": lambda: norm,
            "SBDM": lambda: norm * self.compute_drift(x, t)[1],
            "sigma": lambda: norm * self.compute_sigma_t(t)[0],
            "linear": lambda: norm * (1 - t),
            "decreasing": lambda: 0.25 * (norm * th.cos(math.pi * t) + 1) ** 2,
            "inccreasing-decreasing": lambda: norm * th.sin(math.pi * t) ** 2,
        }
        if form not in forms:
            raise NotImplementedError(f"Diffusion form {form} not implemented")
        return forms[form]()

    def get_score_from_velocity(self, velocity, x, t):
        t = expand_t_like_x(t, x)
        alpha_t, d_alpha_t = self.compute_alpha_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        rev = alpha_t / d_alpha_t
        var = sigma_t ** 2 - rev * d_sigma_t * sigma_t
        return (rev * velocity - x) / var

    def get_noise_from_velocity(self, velocity, x, t):
        t = expand_t_like_x(t, x)
        alpha_t, d_alpha_t = self.compute_alpha_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        rev = alpha_t / d_alpha_t
        return (rev * velocity - x) / (rev * d_sigma_t - sigma_t)

    def get_velocity_from_score(self, score, x, t):
        t = expand_t_like_x(t, x)
        drift, var = self.compute_drift(x, t)
        return var * score - drift

    def compute_mu_t(self, t, x0, x1):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[0] * x1 + self.compute_sigma_t(t)[0] * x0

    def compute_xt(self, t, x0, x1):
        return self.compute_mu_t(t, x0, x1)

    def compute_ut(self, t, x0, x1, xt):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[1] * x1 + self.compute_sigma_t(t)[1] * x0

    def plan(self, t, x0, x1):
        xt = self.compute_xt(t, x0, x1)
        return t, xt, self.compute_ut(t, x0, x1, xt)


class VPCPlan(ICPlan):
    """Variance-preserving path (path.py:139-175)."""

    def __init__(self, sigma_min=0.1, sigma_max=20.0):
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def log_mean_coeff(self, t):
        return -0.25 * ((1 - t) ** 2) * (self.sigma_max - self.sigma_min) - 0.5 * (1 - t) * self.sigma_min

    def d_log_mean_coeff(self, t):
        return 0.5 * (1 - t) * (self.sigma_max - self.sigma_min) + 0.5 * self.sigma_min

    def compute_alpha_t(self, t):
        a = th.exp(self.log_mean_coeff(t))
        return a, a * self.d_log_mean_coeff(t)

    def compute_sigma_t(self, t):
        p = 2 * self.log_mean_coeff(t)
        s = th.sqrt(1 - th.exp(p))
        return s, th.exp(p) * (2 * self.d_log_mean_coeff(t)) / (-2 * s)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return self.d_log_mean_coeff(t)

    def compute_drift(self, x, t):
        t = expand_t_like_x(t, x)
        beta_t = self.sigma_min + (1 - t) * (self.sigma_max - self.sigma_min)
        return -0.5 * beta_t * x, beta_t / 2


class GVPCPlan(ICPlan):
    """Trigonometric path (path.py:178-198)."""

    def compute_alpha_t(self, t):
        return th.sin(t * math.pi / 2), math.pi / 2 * th.cos(t * math.pi / 2)

    def compute_sigma_t(self, t):
        return th.cos(t * math.pi / 2), -math.pi / 2 * th.sin(t * math.pi / 2)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return math.pi / (2 * th.tan(t * math.pi / 2))
