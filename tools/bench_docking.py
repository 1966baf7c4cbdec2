#!/usr/bin/env python3
"""Wall time of the tokenizer's docking functions (decode_to_images, _encode) for 64 images, f32 and bf16 activations.
    python tools/bench_docking.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ldmae_amd.tokenizer import models_mae
m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=True, no_cls=True, kl_loss_weight=True, smooth_output=True, img_size=256).cuda().eval()
z = torch.randn(64, 16, 32, 32, device="cuda")
x = torch.rand(64, 3, 256, 256, device="cuda") * 2 - 1
for tag, prec in (("f32", None), ("bf16", torch.bfloat16)):
    m.set_precision(prec)
    for _ in range(2):
        im = m.decode_to_images(z); lat = m._encode(x)
    torch.cuda.synchronize()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record()
    for _ in range(5): im = m.decode_to_images(z)
    b.record()
    for _ in range(5):
        with torch.no_grad(): lat = m._encode(x)
    c.record(); torch.cuda.synchronize()
    print(f"{tag}: decode_to_images(64 latents) {a.elapsed_time(b) / 5:.2f} ms   _encode(64 images) {b.elapsed_time(c) / 5:.2f} ms")
