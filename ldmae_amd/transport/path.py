"""Coupling plan x_t = alpha_t x1 + sigma_t x0 (reference: LDMAE/transport/path.py).

Only the linear (rectified-flow) plan of the shipped configuration is implemented (configs/imagenet/lightningdit_b_vmae_f8d16_cfg.yaml:
path_type Linear, prediction velocity); the reference's VP / GVP plans and the score / noise conversions are outside the hot-path scope
 and raise in ``create_transport``."""
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
