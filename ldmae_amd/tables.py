"""Host-side, init-time position tables (computed once on the CPU, like the reference does).

* sincos_2d  : frozen 2-D sin-cos `pos_embed` (LDMAE/models/lightningdit.py:444-491 with float64 omega;
               LDMAE/tokenizer/util/pos_embed.py:20-67 with float32 omega).
* rope_2d    : `freqs_cos` / `freqs_sin` buffers of VisionRotaryEmbeddingFast (LDMAE/models/pos_embed.py:96-133).
"""
import numpy as np
import torch


def _axis_table(dim: int, coord: np.ndarray, omega_dtype) -> np.ndarray:
    omega = np.arange(dim // 2, dtype=omega_dtype)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    ang = np.einsum("m,d->md", coord.reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_2d(embed_dim: int, grid_size: int, omega_dtype=np.float64, cls_token: bool = False) -> np.ndarray:
    """[grid*grid (+1), embed_dim]; first half of the channels encodes the column (w) coordinate."""
    assert embed_dim % 4 == 0
    ax = np.arange(grid_size, dtype=np.float32)
    col, row = np.meshgrid(ax, ax)
    tab = np.concatenate([_axis_table(embed_dim // 2, col, omega_dtype), _axis_table(embed_dim // 2, row, omega_dtype)], axis=1)
    if cls_token:
        tab = np.concatenate([np.zeros([1, embed_dim]), tab], axis=0)
    return tab


def rope_2d(dim: int, grid: int, theta: float = 10000.0):
    """cos/sin [grid*grid, 2*dim]: channels [0,dim) rotate with the row index, [dim,2dim) with the column
    index; every frequency is repeated for the (even, odd) pair it rotates."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    pos = torch.arange(grid) / grid * grid
    ang = (pos[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)          # [grid, dim]
    full = torch.cat([ang[:, None, :].expand(grid, grid, dim), ang[None, :, :].expand(grid, grid, dim)], dim=-1)
    full = full.reshape(grid * grid, 2 * dim)
    return full.cos().contiguous(), full.sin().contiguous()
