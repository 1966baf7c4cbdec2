#!/usr/bin/env python3
"""End-to-end sanity at the bench size: the bench's B/1 model, optimizer and transport (bench.build) trained for N steps on a FIXED set of 1024
synthetic latents (4 batches of 256, cycled) with a learning rate of 1e-3 -- the flow-matching loss has to fall steadily as the model fits
the set, with every kernel of the step (static-shift attention, fused epilogues, direct weight gradients, AdamW + EMA) in the loop.
    python tools/train_fit.py [steps=400]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model, opt, reducer, transport = bench.build(dev, 256)
opt.lr = 1e-3
gen = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randn(256, 16, 32, 32, device=dev, generator=gen) for _ in range(4)]
ys = [torch.randint(0, 1000, (256,), device=dev, generator=gen) for _ in range(4)]
t0, hist = time.time(), []
for it in range(steps):
    torch.manual_seed(1000 + it % 8)          # the transport's t / noise draws repeat with period 8: 32 distinct (batch, draw) pairs to fit
    np.random.seed(1000 + it % 8)             # (the lognormal timesteps come from numpy's global generator, as in the reference)
    loss = bench.train_step(model, opt, reducer, transport, xs[it % 4], ys[it % 4])
    if it % 25 == 0 or it == steps - 1:
        hist.append((it, float(loss.detach())))
        print(f"step {it:4d}  loss {hist[-1][1]:.4f}  ({time.time() - t0:.0f} s)", flush=True)
assert all(l == l and l < 1e4 for _, l in hist), "loss is not finite"
print("first / last loss:", hist[0][1], hist[-1][1], "-> falling" if hist[-1][1] < 0.9 * hist[0][1] else "-> NOT falling")
