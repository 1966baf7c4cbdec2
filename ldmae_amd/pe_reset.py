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
    """pe_reset.py:20-77, same keywords.  Returns the path of the new checkpoint."""
    if gradual_resol or pred_with_conv:
        raise NotImplementedError("gradual_resol / pred_with_conv tokenizers are not built here (as vmae_pretrain.py)")
    ckpt_all = torch.load(chkpt_path, map_location="cpu", weights_only=False)
    ckpt = ckpt_all["model"]
    # only the grid of the target model is needed: (input_size / patch)^2 positions
    model = getattr(models_mae, arch)(ldmae_mode=ldmae_mode, no_cls=no_cls, img_size=input_size, smooth_output=smooth_output, kl_loss_weight=kl_loss_weight)
    if model.pos_embed.shape[1] != ckpt["pos_embed"].shape[1]:
        new_size = int(model.pos_embed.shape[1] ** 0.5)
        ckpt["pos_embed"] = resize_pos_embed(ckpt["pos_embed"], new_size)
        ckpt["decoder_pos_embed"] = resize_pos_embed(ckpt["decoder_pos_embed"], new_size)
    if modify_dec_pred:                          # :64-66
        ckpt["decoder_pred.linear_pred.bias"] = ckpt.pop("decoder_pred.bias")
        ckpt["decoder_pred.linear_pred.weight"] = ckpt.pop("decoder_pred.weight")
    if use_initialized_pe:                       # :69-71
        ckpt.pop("pos_embed", None)
        ckpt.pop("decoder_pos_embed", None)
    ckpt_all["model"] = ckpt
    save_path = os.path.splitext(chkpt_path)[0] + "_pe.pth"
    torch.save(ckpt_all, save_path)
    print(f"[+] Saved adjusted checkpoint -> {save_path}")
    return save_path


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
