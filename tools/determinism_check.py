#!/usr/bin/env python3
"""Three full train steps of the bench model (B/1, bs 256, bf16) from fixed seeds -- torch AND numpy: the transport draws its lognormal
timesteps from numpy's global generator, as the reference does -- printing loss, gradient-slab sum and parameter sum at full precision.
Run it twice (two processes, or two boxes): the lines must be identical, digit for digit (fixed reduction orders everywhere, no
atomics on the product path, no read of uninitialised memory).
    python tools/determinism_check.py [batch=256]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model, opt, reducer, transport = bench.build(dev, BS)
gen = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(BS, 16, 32, 32, device=dev, generator=gen)
y = torch.randint(0, 1000, (BS,), device=dev, generator=gen)
print("x", float(x.double().sum()), "params", float(opt.flat.params.double().sum()))
for it in range(3):
    torch.manual_seed(1000 + it)
    import numpy as np; np.random.seed(1000 + it)
    loss = bench.train_step(model, opt, reducer, transport, x, y)
    print(it, repr(float(loss.detach())), "grads", repr(float(opt.flat.grads.double().sum())), "params", repr(float(opt.flat.params.double().sum())))
