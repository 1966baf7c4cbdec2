#!/usr/bin/env python3
"""Sweep of LightningDiT constructor / input geometries through one forward + backward (f32 and bf16): finds shapes the kernels refuse.
    python tools/probe_configs.py"""
import os, sys, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd.models.lightningdit import LightningDiT
base = dict(hidden_size=192, depth=1, num_heads=3, num_classes=10, use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True)
cases = []
for inp, p, c in ((64, 1, 16), (12, 1, 16), (16, 2, 4), (8, 1, 4), (8, 1, 32), (16, 2, 32), (32, 4, 16), (10, 2, 16), (8, 1, 3)):
    cases.append((f"input {inp} patch {p} chans {c}", dict(base, input_size=inp, patch_size=p, in_channels=c), 2))
for B in (1, 3, 5, 7, 9):
    cases.append((f"batch {B}", dict(base, input_size=8, patch_size=1, in_channels=16), B))
for D, H in ((256, 4), (384, 6), (128, 2), (64, 1), (320, 5), (96, 3), (1152, 16), (768, 6), (512, 4)):
    cases.append((f"hidden {D} heads {H} (head dim {D // H})", dict(base, input_size=8, patch_size=1, in_channels=16, hidden_size=D, num_heads=H), 2))
cases.append(("learn_sigma + checkpoint", dict(base, input_size=8, patch_size=1, in_channels=16, learn_sigma=True, use_checkpoint=True), 2))
cases.append(("mlp_ratio 2", dict(base, input_size=8, patch_size=1, in_channels=16, mlp_ratio=2.0), 2))
cases.append(("mlp_ratio 3.5", dict(base, input_size=8, patch_size=1, in_channels=16, mlp_ratio=3.5), 2))
bad = 0
for name, kw, B in cases:
    for prec in (torch.float32, torch.bfloat16):
        try:
            torch.manual_seed(0)
            m = LightningDiT(**kw).cuda().train()
            with torch.no_grad():
                for n, p_ in m.named_parameters():
                    if "adaLN" in n or n.startswith("final_layer.linear"):
                        p_.copy_(torch.randn_like(p_) * 0.02)
            m.set_precision(prec)
            s = kw["input_size"]
            x = torch.randn(B, kw["in_channels"], s, s, device="cuda")
            out = m(x, torch.rand(B, device="cuda"), torch.randint(0, 10, (B,), device="cuda"))
            out.square().mean().backward()
            ok = bool(torch.isfinite(out).all()) and all(torch.isfinite(p_.grad).all() for p_ in m.parameters() if p_.grad is not None)
            print(f"{'ok  ' if ok else 'NAN '} {name:40s} {str(prec)[6:]:9s} out norm {float(out.detach().norm()):.4f}", flush=True)
            bad += not ok
        except Exception as e:
            bad += 1
            print(f"FAIL {name:40s} {str(prec)[6:]:9s} {type(e).__name__}: {str(e)[:150]}", flush=True)
print("failures:", bad)
