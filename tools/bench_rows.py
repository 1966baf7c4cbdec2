#!/usr/bin/env python3
"""The HBM-bound row kernels of the DiT-B/1 train step at bs = 256 (M = 262144 rows of 768): time and achieved TB/s on their algorithmic bytes,
for whatever library LDMAE_HIP_LIB names (A/B of two builds: run twice in one gpurun call).   python tools/bench_rows.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops

M, D, T, B, H, hd = 262144, 768, 1024, 256, 12, 64
g = torch.Generator(device="cuda").manual_seed(0)
bf = torch.bfloat16
rb = lambda *s: torch.randn(*s, device="cuda", generator=g).to(bf)
rf = lambda *s: torch.randn(*s, device="cuda", generator=g)
dout, y, x, dx, w = rb(M, D), rb(M, D), rf(M, D), rf(M, D), rf(D)
mod, rstd = rf(B, 6 * D) * 0.1, torch.rand(M, device="cuda", generator=g) + 0.5
dmod = torch.empty_like(mod)
qkv = rb(M, 3 * D)
wq, wk = rf(hd), rf(hd)
cos, sin = torch.rand(T, hd, device="cuda", generator=g), torch.rand(T, hd, device="cuda", generator=g)
cases = [
    ("rmsnorm_modulate_bwd_gate", (2 + 4 + 8 + 2 + 2) * M * D, lambda: ops.rmsnorm_modulate_bwd_gate(dout, x, w, mod[:, D:2 * D], rstd, dx, dmod[:, 0:D], dmod[:, D:2 * D], y, mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], T, bf)),
    ("rmsnorm_modulate_bwd     ", (2 + 4 + 8) * M * D, lambda: ops.rmsnorm_modulate_bwd(dout, x, w, mod[:, D:2 * D], rstd, dx, dmod[:, 0:D], dmod[:, D:2 * D], T)),
    ("rmsnorm_modulate_fwd     ", (4 + 2) * M * D, lambda: ops.rmsnorm_modulate_fwd(x, w, mod[:, 0:D], mod[:, D:2 * D], T, bf)),
    ("qknorm_rope_fwd          ", (2 + 2) * M * 2 * D, lambda: ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, T, H, hd, copy_v=False)),
    ("gate_bwd                 ", (4 + 2 + 2) * M * D, lambda: ops.gate_bwd(dx, y, mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], T, bf, with_bias=True)),
]
print(os.path.basename(os.environ.get("LDMAE_HIP_LIB", "libldmae_hip.so")))
for name, nbytes, fn in cases:
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    print(f"  {name}: {best:.3f} ms  {nbytes / best / 1e9:.2f} TB/s on {nbytes / 1e9:.2f} GB (incl. the small reduce launches)")
