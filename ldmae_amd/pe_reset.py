#!/usr/bin/env python3
"""Stage 2 of the tokenizer recipe (`train_ae.sh`): counterpart of the reference's VMAE/pe_reset.py.

Resizes `pos_embed` and `decoder_pos_embed` of a pre-training checkpoint to the grid of the target resolution (bilinear, the same rule the
resume path applies: VMAE/util/misc.py:488-499) and writes `<checkpoint>_pe.pth` beside it.  Host-side only: nothing here touches the GPU.
(The reference's file imports `models_mae.util.pos_embed.resize_pos_embed`, a path that does not exist in its tree; the function it means is
`util/misc.py: resize_pos_embed`, restated in `vmae_pretrain.resize_pos_embed`.)"""
import argparse
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
for p in (_HERE, os.path.dirname(_HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

from ldmae_amd.tokenizer import models_mae                      # noqa: E402
from ldmae_amd.vmae_pretrain import resize_pos_embed            # noqa: E402


def reset_positional_embedding(chkpt_path, arch="mae_for_ldmae_f8d16_prev", *, input_size=256, ldmae_mode=True, no_cls=True, gradual_resol=False,
                               smooth_output=True, pred_with_conv=False, kl_loss_weight=None, use_initialized_pe=False, modify_dec_pred=False):
    """pe_reset.py:20-77, same keywords and the same result.  Returns the path of the new checkpoint."""
    if gradual_resol or pred_with_conv:
        raise NotImplementedError("gradual_resol / pred_with_conv tokenizers are not built here (as vmae_pretrain.py)")
    blob = torch.load(chkpt_path, map_location="cpu", weights_only=False)
    sd = blob["model"]
    # only the GRID of the target model matters: (input_size / patch)^2 positions, read off a freshly built tokenizer of that architecture
    target = getattr(models_mae, arch)(ldmae_mode=ldmae_mode, no_cls=no_cls, img_size=input_size, smooth_output=smooth_output, kl_loss_weight=kl_loss_weight)
    side = int(round(target.pos_embed.shape[1] ** 0.5))
    for key in ("pos_embed", "decoder_pos_embed"):
        if use_initialized_pe:                       # :69-71 -- drop them: the loader then keeps the model's own sin-cos tables
            sd.pop(key, None)
        elif sd["pos_embed"].shape[1] != side * side or sd[key].shape[1] != side * side:      # :58-61 -- both follow the encoder table's grid
            sd[key] = resize_pos_embed(sd[key], side)
    if modify_dec_pred:                              # :64-66 -- the two-layer predictor's key names
        for leaf in ("bias", "weight"):
            sd[f"decoder_pred.linear_pred.{leaf}"] = sd.pop(f"decoder_pred.{leaf}")
    out_path = os.path.splitext(chkpt_path)[0] + "_pe.pth"
    torch.save(blob, out_path)
    print(f"[+] Saved adjusted checkpoint -> {out_path}")
    return out_path


def main(argv=None):
    ap = argparse.ArgumentParser(description="Adjust positional embeddings in an LDMAE checkpoint")
    ap.add_argument("--model_name", default="mae_for_ldmae_f8d16_prev", help="Architecture registered in models_mae")
    # the reference's parser names the flag --chkpt_dir (pe_reset.py:90) while train_ae.sh passes --ckpt_dir: both are taken
    ap.add_argument("--chkpt_dir", "--ckpt_dir", dest="chkpt_dir", required=True, help="Path to the original checkpoint (.pth)")
    ap.add_argument("--input_size", type=int, default=256)
    a = ap.parse_args(argv)
    return reset_positional_embedding(a.chkpt_dir, a.model_name, input_size=a.input_size)


if __name__ == "__main__":
    main()
