"""RMSNorm (reference: LDMAE/models/rmsnorm.py:34-77; only the class the DiT imports, lightningdit.py:24)."""
import torch
from torch import nn

from ldmae_amd import ops


class _RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        shp = x.shape
        x2 = x.float().contiguous().view(-1, shp[-1])
        out, rstd = ops.rmsnorm_modulate_fwd(x2, w, None, None, x2.shape[0], torch.float32, eps)
        ctx.save_for_backward(x2, w, rstd)
        return out.view(shp)

    @staticmethod
    def backward(ctx, g):
        x2, w, rstd = ctx.saved_tensors
        dx = torch.empty_like(x2)
        dw = ops.rmsnorm_modulate_bwd(g.float().contiguous().view_as(x2), x2, w, None, rstd, dx, None, None, x2.shape[0], accumulate=False)
        return dx.view(g.shape), dw, None


class RMSNorm(nn.Module):
    """`x * rsqrt(mean(x^2) + eps) * weight` over the last dimension, statistics in f32."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return _RMSNormFn.apply(x, self.weight, self.eps)
