import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ldmae_amd") else os.getcwd())
from ldmae_amd import _lib, ops
M = 262144
g = torch.Generator(device='cuda').manual_seed(0)
def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / iters
tot = 0
for N, K in [(2304, 768), (768, 768), (4096, 768), (768, 2048)]:
    a = torch.randn(M, N, device='cuda', generator=g).to(torch.bfloat16)
    b = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    t = min(timed(lambda: ops.gemm_tn(a, b)) for _ in range(4)); tot += t
    print(f"TN N={N} K={K}: {t:.3f} ms {2.0 * M * N * K / t / 1e9:.0f} TF/s", end=" | ")
print(f"total {tot:.3f} ms")
