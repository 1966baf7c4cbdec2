"""2-D rotary position tables for the attention kernels (reference: LDMAE/models/pos_embed.py:96-135)."""
import torch
from torch import nn

from ..tables import rope_2d


class VisionRotaryEmbeddingFast(nn.Module):
    """Holds the `freqs_cos` / `freqs_sin` [N, head_dim] buffers (same names / shapes as the reference,
    so checkpoints load).  The rotation itself is fused into ldmae_qknorm_rope_{fwd,bwd}: this module
    is only a table holder and is not callable on its own."""

    def __init__(self, dim, pt_seq_len=16, ft_seq_len=None, custom_freqs=None, freqs_for='lang', theta=10000,
                 max_freq=10, num_freqs=1):
        super().__init__()
        if custom_freqs is not None or freqs_for != 'lang' or (ft_seq_len not in (None, pt_seq_len)):
            raise NotImplementedError("ldmae_amd: only the 'lang' frequencies with ft_seq_len == pt_seq_len are supported "
                                      "(what LightningDiT builds, lightningdit.py:317-323)")
        cos, sin = rope_2d(dim, pt_seq_len, theta)
        self.register_buffer("freqs_cos", cos)
        self.register_buffer("freqs_sin", sin)

    def forward(self, t):
        raise NotImplementedError("RoPE is fused into the attention front-end kernel (ldmae_qknorm_rope_fwd); "
                                  "VisionRotaryEmbeddingFast only holds the tables")
