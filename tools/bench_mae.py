#!/usr/bin/env python3
"""Secondary measurement: VMAE masked-token encoder throughput on one MI355X.
mae_for_ldmae_f8d16_prev(no_cls=True, smooth_output=True, img_size=256), batch 256 of 256x256x3 images, mask_ratio 0.75 -> 256 kept tokens
per image.  forward_encoder only (patch-embed 8x8, random masking, 12 pre-LN blocks of width 192 / head_dim 16, LayerNorm), inference mode.
    python tools/bench_mae.py [--batch 256] [--steps 20] [--bf16]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd.tokenizer import models_mae


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--ratio", type=float, default=0.75)
    ap.add_argument("--mode", default="both", choices=["both", "fp32", "bf16"])
    args = ap.parse_args()
    m = models_mae.mae_for_ldmae_f8d16_prev(ldmae_mode=False, no_cls=True, kl_loss_weight=1e-6, smooth_output=True, img_size=256).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand(args.batch, 3, 256, 256, device="cuda", generator=g) * 2 - 1)
    for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
        if args.mode != "both" and not name.startswith(args.mode):
            continue
        with torch.no_grad(), ctx:
            for _ in range(3):
                lat, mask, ids = m.forward_encoder(x, args.ratio)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                lat, mask, ids = m.forward_encoder(x, args.ratio)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        kept = lat.shape[1]
        print(f"{name:14s}: {dt * 1e3:7.2f} ms / batch of {args.batch}  -> {args.batch / dt:8.0f} images/s, {args.batch * kept / dt / 1e6:6.2f} M kept tokens/s "
              f"(latent {tuple(lat.shape)}, {3.32 * args.batch / dt / 1e3:.1f} TFLOP/s algorithmic at 3.32 GFLOP/sample)")


if __name__ == "__main__":
    main()
