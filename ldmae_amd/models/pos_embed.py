"""2-D rotary position tables for the attention kernels (reference: LDMAE/models/pos_embed.py:96-135)."""
import torch
from torch import nn

from ldmae_amd import ops
from ldmae_amd.tables import rope_2d


class _RopeFn(torch.autograd.Function):
    """t * cos + rotate_half(t) * sin (:135, rotate_half :38-42); the backward is the adjoint rotation."""

    @staticmethod
    def forward(ctx, t, cos, sin):
        ctx.save_for_backward(cos, sin)
        return ops.rope(t, cos, sin)

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        return ops.rope(g.contiguous(), cos, sin, transposed=True), None, None


class VisionRotaryEmbeddingFast(nn.Module):
    """Holds the `freqs_cos` / `freqs_sin` [N, head_dim] buffers (same names / shapes as the reference,
    so checkpoints load).  Inside LightningDiTBlock the rotation is fused into ldmae_qknorm_rope_{fwd,bwd}
    (the block hands the tables to the kernel); called on its own -- `rope(q)` as the reference's Attention.forward does (:72-73) --
    it runs the standalone ldmae_rope kernel on t [..., N, head_dim] (f32 or bf16)."""

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
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        return _RopeFn.apply(t, self.freqs_cos, self.freqs_sin)
