"""Deterministic, name-keyed test weights shared by make_golden.py (which loads
them into the *reference* model) and by the tests (which load them into the
oracle and the HIP path).  Weights are never stored in fixtures: both sides
regenerate them from (name, shape, seed) with torch's CPU generator.
"""
import math
import zlib

import numpy as np
import torch


def det_randn(name: str, shape, seed: int = 0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) % (2 ** 31))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


def det_weights(shapes: dict, seed: int = 0, skip=("pos_embed", "decoder_pos_embed")) -> dict:
    """Non-degenerate values for every parameter: adaLN / final layers are NOT
    zero (zero-init layers make goldens test nothing)."""
    out = {}
    for k, shp in shapes.items():
        if k in skip:
            continue
        z = det_randn(k, shp, seed)
        if k.endswith("norm.weight") or "norm1.weight" in k or "norm2.weight" in k or "norm_final.weight" in k:
            out[k] = 1.0 + 0.1 * z
        elif k.endswith(".bias"):
            out[k] = 0.05 * z
        elif "embedding_table" in k:
            out[k] = 0.5 * z
        elif "adaLN_modulation" in k:
            out[k] = z * (0.5 / math.sqrt(int(np.prod(shp[1:]))))
        else:
            out[k] = z / math.sqrt(int(np.prod(shp[1:])))
    return out


def ref_style_init(shapes: dict, seed: int = 10) -> dict:
    """The reference's init SCHEME (lightningdit.py:340-374: Xavier Linears, zero biases / adaLN / final linear,
    N(0,.02) embedders, unit norms) with name-keyed values -- identical to make_golden.ref_style_init, which
    loads the same numbers into the reference model for the 100-step loss curve."""
    out = {}
    for k, shp in shapes.items():
        if k == "pos_embed":
            continue
        z = det_randn(k, shp, seed)
        if "norm" in k and k.endswith("weight"):
            out[k] = torch.ones(shp)
        elif k.endswith(".bias") or "adaLN_modulation" in k or k.startswith("final_layer.linear"):
            out[k] = torch.zeros(shp)
        elif k.startswith("y_embedder") or k.startswith("t_embedder"):
            out[k] = 0.02 * z
        else:
            fan_out, fan_in = shp[0], int(np.prod(shp[1:]))
            out[k] = z * math.sqrt(2.0 / (fan_in + fan_out))
    return out


# one entry per constructor flag of LightningDiTBlock the shipped imagenet YAML leaves at its default (lightningdit.py:292-296), the reference's
# CelebA-HQ configuration first (configs/celeba_hq/lightningdit_b_vmae_f8d16_cfg.yaml:13-35: use_qknorm false, num_classes 1).  Shared with the tests.
DIT_FLAG_VARIANTS = {
    "noqk": dict(use_qknorm=False, num_classes=1),
    "woshift": dict(wo_shift=True),
    "norope": dict(use_rope=False),
    "ln": dict(use_rmsnorm=False),
    "mlp": dict(use_swiglu=False),
    "plain": dict(use_qknorm=False, use_swiglu=False, use_rope=False, use_rmsnorm=False, wo_shift=True),
}
