#!/usr/bin/env python3
"""bf16 TN (weight-gradient) GEMM variants on the LightningDiT-B/1 bs=256 shapes, with and without the fused bias gradient."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))   # A/B knobs live in the diagnostic build only (make -C ldmae_amd/csrc diag)
from ldmae_amd import _lib, ops
lib = _lib.load()
M = 262144
g = torch.Generator(device='cuda').manual_seed(0)
def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / iters
for N, K in [(2304, 768), (768, 768), (4096, 768), (768, 2048)]:
    a = torch.randn(M, N, device='cuda', generator=g).to(torch.bfloat16)
    b = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    ref = a[:, :64].float().T @ b[:, :64].float()
    for st, name in [(0, 'target256'), (512, 'target512'), (192, 'target192'), (128, 'target128')]:
        lib.ldmae_tune(9, st)
        out, db = ops.gemm_tn(a, b, with_bias=True)
        err = float((out[:64, :64] - ref).norm() / ref.norm())
        t = min(timed(lambda: ops.gemm_tn(a, b, with_bias=True)) for _ in range(3))
        t0 = min(timed(lambda: ops.gemm_tn(a, b, with_bias=False)) for _ in range(3))
        print(f"TN N={N} K={K} {name:9s}: with bias {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f} TF/s | no bias {t0:.3f} ms {2.0 * M * N * K / t0 / 1e9:.0f} TF/s err={err:.1e}")
lib.ldmae_tune(9, 0)
