#!/usr/bin/env python3
"""A few calls of the tokenizer's docking functions at batch 256 for rocprofv3 --kernel-trace --stats (which kernels carry them).
    python tools/docking_profile.py f32|bf16 encode|decode"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd.tokenizer import models_mae
prec, what = sys.argv[1], sys.argv[2]
m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=256).cuda().eval()
m.set_precision(torch.bfloat16 if prec == "bf16" else None)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand(256, 3, 256, 256, device="cuda", generator=g) * 2 - 1
z = torch.randn(256, 16, 32, 32, device="cuda", generator=g)
with torch.no_grad():
    for _ in range(3):
        out = m._encode(x) if what == "encode" else m.decode_to_images(z)
torch.cuda.synchronize()
