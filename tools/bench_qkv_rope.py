#!/usr/bin/env python3
"""A/B of the qkv front end of the LightningDiT-B/1 block at bs 256 (M 262144, K 768, 12 heads of 64, bf16): ldmae_gemm_nt + ldmae_qknorm_rope_fwd (the
pair) against ldmae_gemm_nt_qkv_rope (QK-norm / RoPE in the GEMM's epilogue), interleaved rounds in one process; and the forward-only form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops  # noqa: E402


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    B, N, H, hd, K = 256, 1024, 12, 64, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(B * N, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(3 * H * hd, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(3 * H * hd, device="cuda", generator=g) * 0.1
    wq, wk = 1 + 0.1 * torch.randn(hd, device="cuda", generator=g), 1 + 0.1 * torch.randn(hd, device="cuda", generator=g)
    cos, sin = torch.rand(N, hd, device="cuda", generator=g), torch.rand(N, hd, device="cuda", generator=g)

    def pair():
        qkv = ops.gemm_nt(a, w, bias)
        return (qkv,) + ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, 1e-6, copy_v=False)[:2]
    fused = lambda: ops.gemm_nt_qkv_rope(a, w, bias, wq, wk, cos, sin, B, N, H, hd, 1e-6)                        # noqa: E731
    fwd_only = lambda: ops.gemm_nt_qkv_rope(a, w, bias, wq, wk, cos, sin, B, N, H, hd, 1e-6, store_raw_qk=False)  # noqa: E731
    gemm = lambda: ops.gemm_nt(a, w, bias)                                                                         # noqa: E731
    p, f = pair(), fused()
    print("bitwise equal (qkv, q2, k2):", [bool(torch.equal(x, y)) for x, y in zip(p, f)])
    for r in range(4):
        print(f"round {r}: GEMM alone {timed(gemm):.3f} ms | pair {timed(pair):.3f} ms | fused {timed(fused):.3f} ms | fused, forward-only {timed(fwd_only):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
