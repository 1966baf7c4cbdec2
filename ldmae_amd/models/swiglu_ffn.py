"""SwiGLU feed-forward (reference: LDMAE/models/swiglu_ffn.py:15-36)."""
from typing import Callable, Optional

import torch
from torch import Tensor, nn

from ldmae_amd import ops


class _SwiGLUFn(torch.autograd.Function):
    """w3(silu(x1) * x2), [x1 | x2] = w12(x); activations in x's dtype (f32 or bf16)."""

    @staticmethod
    def forward(ctx, x, w12, b12, w3, b3):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        dt = x2.dtype
        W12, W12T = (w12, ops.cast_weight(w12, dt, True, False)[1]) if dt == torch.float32 else ops.cast_weight(w12, dt)
        W3, W3T = (w3, ops.cast_weight(w3, dt, True, False)[1]) if dt == torch.float32 else ops.cast_weight(w3, dt)
        h12, hid = ops.gemm_nt_swiglu(x2, W12, b12)
        out = ops.gemm_nt(hid, W3, b3)
        ctx.save_for_backward(x2, h12, hid, W12T, W3T)
        return out.view(*shp[:-1], -1)

    @staticmethod
    def backward(ctx, g):
        x2, h12, hid, W12T, W3T = ctx.saved_tensors
        g2 = g.contiguous().view(x2.shape[0], -1)
        dw3, db3 = ops.gemm_tn(g2, hid, with_bias=True)
        dh12 = ops.gemm_nt_swiglu_bwd(g2, W3T, h12)
        dw12, db12 = ops.gemm_tn(dh12, x2, with_bias=True)
        return ops.gemm_nt(dh12, W12T).view(*g.shape[:-1], -1), dw12, db12, dw3, db3


class SwiGLUFFN(nn.Module):
    def __init__(self, in_features: int, hidden_features: Optional[int] = None, out_features: Optional[int] = None,
                 act_layer: Callable[..., nn.Module] = None, drop: float = 0.0, bias: bool = True) -> None:
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x: Tensor) -> Tensor:
        return _SwiGLUFn.apply(x, self.w12.weight, self.w12.bias, self.w3.weight, self.w3.bias)
