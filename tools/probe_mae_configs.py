#!/usr/bin/env python3
"""Sweep of the tokenizer registry / image geometries through one pre-training step (f32, bf16, fp16 autocast) and the docking calls: finds shapes the
kernels refuse.     python tools/probe_mae_configs.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd.tokenizer import models_mae as mm
names = ["mae_for_ldmae", "mae_for_ldmae_f8d32", "mae_for_ldmae_f8d16_prev", "mae_for_ldmae_f8d16_small", "mae_for_ldmae_f8d16_asym_small", "mae_for_ldmae_f8d16_prev_large",
         "mae_for_ldmae_f8d16", "mae_for_ldmae_f8d16_flexible", "mae_for_ldmae_f16d32", "mae_for_ldmae_f16d32_large", "mae_for_ldmae_f8d32_flexible", "mae_for_ldmae_16d",
         "mae_vit_base_patch16_dec512d8b", "mae_vit_base_patch16_dec128d8b", "mae_vit_large_patch16_dec512d8b", "mae_vit_huge_patch14_dec512d8b"]
cases = [(n, dict(), None) for n in names]
for size in (96, 120, 136, 224):
    cases.append(("mae_for_ldmae_f8d16_prev", dict(), size))
bad = 0
for n, kw, size in cases:
    f = getattr(mm, n)
    for prec in ("fp32", "bf16", "fp16"):
        try:
            torch.manual_seed(0)
            try:
                probe = f(no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=64)
                p = probe.patch_embed.patch_size[0]
                S = size or (p * 8 if p != 14 else 112)
                m = f(no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=S)
            except TypeError:                    # constructors that fix img_size themselves (as in the reference)
                m = f(no_cls=True, kl_loss_weight=1e-6, smooth_output=True)
                S = m.img_size
            m.blocks, m.decoder_blocks = m.blocks[:1], m.decoder_blocks[:1]          # depth 1: shapes, not depth, are what is probed
            m = m.cuda().train()
            x = (torch.rand(3, 3, S, S, device="cuda") * 2 - 1)
            with torch.autocast("cuda", dtype=torch.float16 if prec == "fp16" else torch.bfloat16, enabled=prec != "fp32"):
                loss = m(x, mask_ratio=0.75, visible_loss_ratio=0.5)[0]
            loss.backward()
            ok = bool(torch.isfinite(loss)) and all(torch.isfinite(q.grad).all() for q in m.parameters() if q.grad is not None)
            m.eval()
            with torch.no_grad():
                rec = m.decode(m._encode(x)[:, :m.latent_dim]).sample
            ok = ok and rec.shape == x.shape and bool(torch.isfinite(rec).all())
            print(f"{'ok  ' if ok else 'NAN '} {n:34s} img {S:4d} {prec:5s} loss {float(loss):.4f}", flush=True)
            bad += not ok
        except Exception as e:
            bad += 1
            print(f"FAIL {n:34s} img {size} {prec:5s} {type(e).__name__}: {str(e)[:170]}", flush=True)
print("failures:", bad)
