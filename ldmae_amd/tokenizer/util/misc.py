"""Latent posterior used by the tokenizer and the latent dataset (reference: LDMAE/tokenizer/util/misc.py:74-128).
Tiny elementwise math on [B, 2*latent, h, w] moments: stays in torch."""
from typing import Optional, Tuple

import numpy as np
import torch


class DiagonalGaussianDistribution(object):
    """The reference has TWO versions of this class: LDMAE/tokenizer/util/misc.py:74-128 (what the LDMAE drivers import; `kl()` with the mean^2
    term) and VMAE/util/misc.py:74-140 of the pre-training tree (`fixed_std` argument; `kl()` = 0.5 sum(var / s^2 - 1 - logvar + log s^2) with it,
    and WITHOUT it the variance-only 0.5 sum(var - 1 - logvar): :103-125).  `pretrain_tree=True` selects the second (the pre-training step sets
    it: MaskedAutoencoderViT.kl_form); sample / mode / nll are the same in both."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False, fixed_std=None, pretrain_tree: bool = False):
        self.fixed_std, self.pretrain_tree = fixed_std, bool(pretrain_tree or fixed_std is not None)
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def kl(self, other: "DiagonalGaussianDistribution" = None) -> torch.Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        dims = list(range(1, self.mean.dim()))
        if self.pretrain_tree and self.fixed_std is not None:                       # VMAE/util/misc.py:105-116
            fixed_var = torch.tensor(self.fixed_std) ** 2
            return 0.5 * torch.sum(self.var / fixed_var - 1.0 - self.logvar + torch.log(fixed_var), dim=dims)
        if other is None and self.pretrain_tree:                                    # VMAE/util/misc.py:118-125 (no mean^2 term in that tree)
            return 0.5 * torch.sum(self.var - 1.0 - self.logvar, dim=dims)
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=dims)
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0 - self.logvar + other.logvar,
                               dim=dims)

    def nll(self, sample: torch.Tensor, dims: Tuple[int, ...] = (1, 2, 3)) -> torch.Tensor:
        if self.deterministic:
            return torch.Tensor([0.0])
        return 0.5 * torch.sum(np.log(2.0 * np.pi) + self.logvar + torch.pow(sample - self.mean, 2) / self.var, dim=list(dims))

    def mode(self) -> torch.Tensor:
        return self.mean
