"""2-D sin-cos position table for the VMAE (reference: LDMAE/tokenizer/util/pos_embed.py:20-67, float32 omega)."""
import numpy as np

from ldmae_amd.tables import sincos_2d


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    return sincos_2d(embed_dim, grid_size, np.float32, cls_token=cls_token)
