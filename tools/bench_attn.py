#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels on the LightningDiT-B/1 shape (H=12, N=1024, hd=64, bf16, random data), with
torch SDPA timed beside it as the known-good reference on the same device."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops  # noqa: E402


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=256)
    ap.add_argument("--ref", action="store_true")
    ap.add_argument("--fused", action="store_true")
    args = ap.parse_args()
    B, H, N, hd = args.b, 12, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = (torch.randn(B, H, N, hd, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    do = torch.randn(B, N, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    scale = hd ** -0.5
    fl = 4.0 * B * H * N * N * hd
    o, lse = ops.attention_fwd(q, k, v, scale)
    ref = torch.nn.functional.scaled_dot_product_attention(q[:2].float(), k[:2].float(), v[:2].float()).transpose(1, 2).reshape(2, N, H * hd)
    print("fwd rel err vs f32 sdpa:", float((o[:2].float() - ref).norm() / ref.norm()))
    for r in range(3):
        t = timed(lambda: ops.attention_fwd(q, k, v, scale))
        tb = timed(lambda: ops.attention_bwd(q, k, v, o, do, lse, scale))
        print(f"round {r}: fwd {t:.3f} ms {fl / t / 1e9:7.1f} TF/s | bwd {tb:.3f} ms {2.5 * fl / tb / 1e9:7.1f} TF/s (algorithmic 2.5x fwd)")
    if args.fused:
        # LightningDiT block form: v in the packed qkv, QK-norm / RoPE backward either as its own pass or inside the attention epilogues
        qkv = torch.randn(B, N, 3, H, hd, device="cuda", generator=g).to(torch.bfloat16)
        wq, wk = torch.ones(hd, device="cuda"), torch.ones(hd, device="cuda")
        cos, sin = torch.rand(N, hd, device="cuda", generator=g), torch.rand(N, hd, device="cuda", generator=g)
        q2, k2, _ = ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, copy_v=False)
        o2, lse2 = ops.attention_fwd_pv(q2, k2, qkv, scale)

        def two_pass():
            dq, dk, dqkv = ops.attention_bwd_pv(q2, k2, qkv, o2, do, lse2, scale)
            return ops.qknorm_rope_bwd(dq, dk, None, qkv, wq, wk, cos, sin, B, N, H, hd, with_bias=True, dqkv=dqkv)
        for r in range(3):
            ta = timed(two_pass)
            tb = timed(lambda: ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin))
            print(f"round {r}: attention_bwd_pv + qknorm_rope_bwd {ta:.3f} ms | attention_bwd_pv_qknorm {tb:.3f} ms")
    if args.ref:
        qq, kk, vv = (x.clone().requires_grad_(True) for x in (q, k, v))
        t = timed(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv))
        out = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv)
        gg = torch.randn_like(out)
        tb = timed(lambda: torch.autograd.grad(out, (qq, kk, vv), gg, retain_graph=True))
        print(f"torch SDPA: fwd {t:.3f} ms {fl / t / 1e9:7.1f} TF/s | bwd {tb:.3f} ms {2.5 * fl / tb / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
