#!/usr/bin/env python3
"""Diagnostic ablations of the bf16 NT GEMM (full / no-MFMA / no-loads builds of the same kernel, and tiny-K launches that
isolate the per-tile fixed cost: launch + pipeline fill + epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for N, K in [(2304, 768), (768, 64), (768, 768)]:
    a = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).to(torch.bfloat16)
    for dbg, name in [(0, 'full'), (21, 'no-mfma'), (22, 'no-loads'), (23, 'no-epilog')]:
        lib.ldmae_tune(0, dbg)
        t = min(timed(lambda: ops.gemm_nt(a, w, None)) for _ in range(3))
        tiles_per_cu = (M // 256) * ((N + 255) // 256) / 256
        print(f"N={N} K={K} {name:9s} {t:.3f} ms  ({2.0 * M * N * K / t / 1e9:.0f} TF/s-equivalent, {t * 1e3 / tiles_per_cu:.2f} us per tile per CU)")
    lib.ldmae_tune(0, 0)
