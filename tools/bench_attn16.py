#!/usr/bin/env python3
"""A/B of the head-dim-16 attention backward (the VMAE heads at 256 images): the one-kernel form (attn_bwd_fused16_kernel, diagnostic
build only, tune key 21; ablations tune key 22) against the shipped dQ + dK/dV kernel pair, same process, same data.
    make -C ldmae_amd/csrc diag && LDMAE_HIP_LIB=ldmae_amd/libldmae_hip_diag.so python tools/bench_attn16.py [--f16]
Evidence: profiles/r05_attn16_onepass.txt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import _lib, ops  # noqa: E402


def ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    lib = _lib.load()
    if not hasattr(lib, "ldmae_tune"):
        sys.exit("needs the diagnostic build: LDMAE_HIP_LIB=ldmae_amd/libldmae_hip_diag.so")
    dtype = torch.float16 if "--f16" in sys.argv else torch.bfloat16
    B, H, hd = 256, 12, 16
    if "--prefetch" in sys.argv:
        # round 6: operand-fragment prefetch (template parameter PF of the backward pair; tune key 25: 1 = PF 1, 3 = none; shipped at head_dim 16: none) -- timing and bitwise equality
        for N in (1024, 256):
            g = torch.Generator().manual_seed(0)
            qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(dtype).cuda()
            do = torch.randn(B, N, H * hd, generator=g).to(dtype).cuda()
            o, lse = ops.attention_fwd_qkv(qkv, B, N, H, hd, hd ** -0.5)
            bwd = lambda: ops.attention_bwd_qkv(qkv, o, do, lse, B, N, H, hd, hd ** -0.5)      # noqa: E731
            res = {1: [], 3: []}
            for mode in (3, 1, 3, 1, 3, 1):
                lib.ldmae_tune(25, mode)
                res[mode].append(ms(bwd))
            lib.ldmae_tune(25, 3)
            a_ = bwd()
            lib.ldmae_tune(25, 1)
            b_ = bwd()
            lib.ldmae_tune(25, 0)
            print(f"N {N:5d} {str(dtype)[6:]}: backward pair without prefetch {min(res[3]):.3f} ms ({res[3]}) | PF 1 {min(res[1]):.3f} ms ({res[1]}) | bitwise equal {torch.equal(a_, b_)}", flush=True)
        return
    for N in (1024, 256, 512, 768):
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(dtype).cuda()
        do = torch.randn(B, N, H * hd, generator=g).to(dtype).cuda()
        o, lse = ops.attention_fwd_qkv(qkv, B, N, H, hd, hd ** -0.5)
        fwd = ms(lambda: ops.attention_fwd_qkv(qkv, B, N, H, hd, hd ** -0.5))
        bwd = lambda: ops.attention_bwd_qkv(qkv, o, do, lse, B, N, H, hd, hd ** -0.5)      # noqa: E731
        res = {}
        for mode in (1, 0, 1, 0):
            lib.ldmae_tune(21, mode)
            res.setdefault(mode, []).append(ms(bwd))
        lib.ldmae_tune(21, 1)
        one, again = bwd(), bwd()
        lib.ldmae_tune(21, 0)
        pair = bwd()
        v = lambda t: t.view(B, N, 3, H, hd)      # noqa: E731
        print(f"N {N:5d} {str(dtype)[6:]}: forward {fwd:.3f} ms | backward, one kernel {min(res[1]):.3f} ms | dQ + dK/dV pair {min(res[0]):.3f} ms | "
              f"one kernel reproducible {torch.equal(one, again)}, dK/dV bitwise equal to the pair {torch.equal(v(one)[:, :, 1:], v(pair)[:, :, 1:])}, "
              f"dQ rel diff {rel(v(one)[:, :, 0], v(pair)[:, :, 0]):.2e}", flush=True)
        if N == 1024:
            lib.ldmae_tune(21, 1)
            for dbg, what in ((1, "no dQ slabs / reduce / second barrier"), (2, "no dS image, no dQ product"), (3, "1 + 2"), (4, "no exponential"),
                              (8, "no accumulator-initialising LDS reads"), (7, "1 + 2 + 4"), (15, "1 + 2 + 4 + 8")):
                lib.ldmae_tune(22, dbg)
                print(f"        ablation {dbg:2d} ({what}): {ms(bwd):.3f} ms", flush=True)
            lib.ldmae_tune(22, 0)
            lib.ldmae_tune(21, 0)


if __name__ == "__main__":
    main()
