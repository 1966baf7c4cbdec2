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


def prefetch_ab(B):
    """Round 6: each MFMA of the backward kernels took its LDS operand fragment right in front of it behind `s_waitcnt lgkmcnt(0)`; PF = n requests it n
    MFMAs ahead (csrc/attention.hip, template parameter PF).  Same arithmetic in the same order: outputs must be bitwise equal."""
    from ldmae_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, "ldmae_tune"):
        sys.exit("needs the diagnostic build: LDMAE_HIP_LIB=ldmae_amd/libldmae_hip_diag.so")
    H, N, hd = 12, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    scale = hd ** -0.5
    qkv = torch.randn(B, N, 3, H, hd, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B, N, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    wq, wk = 1 + 0.1 * torch.randn(hd, device="cuda", generator=g), 1 + 0.1 * torch.randn(hd, device="cuda", generator=g)
    cos, sin = torch.rand(N, hd, device="cuda", generator=g), torch.rand(N, hd, device="cuda", generator=g)
    q2, k2, _ = ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, copy_v=False)
    o2, lse2 = ops.attention_fwd_pv(q2, k2, qkv, scale)
    run = lambda: ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin)      # noqa: E731
    # tune values: 3 = no prefetch, 1 / 2 = that depth, 0 = the shipped default; dK/dV only: 4 / 5 = depth 1 / 2 with the two 32-row halves software-pipelined (PIPE)
    combos = [(3, 3), (1, 3), (3, 1), (1, 1), (2, 3), (3, 2), (2, 2), (4, 1), (5, 1), (0, 0)]
    base = None
    for kv, kq in combos:
        lib.ldmae_tune(23, kv); lib.ldmae_tune(24, kq)
        out = run()
        if base is None:
            base = out
        print(f"dK/dV key {kv}, dQ key {kq}: bitwise equal to the form without prefetch:", all(torch.equal(x, y) for x, y in zip(out, base)))
    for r in range(3):
        line = []
        for kv, kq in combos:
            lib.ldmae_tune(23, kv); lib.ldmae_tune(24, kq)
            line.append(f"({kv},{kq}) {timed(run, 8):.3f}")
        print(f"round {r} [ms, (dK/dV key, dQ key); 3 = no prefetch, 0 = shipped]: " + " | ".join(line))
    lib.ldmae_tune(23, 0); lib.ldmae_tune(24, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=256)
    ap.add_argument("--ref", action="store_true")
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--prefetch", action="store_true", help="diagnostic build: operand-fragment prefetch depth of the fused dK/dV (tune key 23) and dQ (24) kernels, "
                                                            "every combination against the shipped form: timing (interleaved rounds) and bitwise equality")
    args = ap.parse_args()
    if args.prefetch:
        return prefetch_ab(args.b)
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
        from ldmae_amd import _lib
        lib = _lib.load()
        diag = hasattr(lib, "ldmae_tune")           # diagnostic build: key 17 = 1 = the one-pass backward kernel instead of the two kernels (dQ, then dK/dV)
        for r in range(3):
            ta = timed(two_pass)
            tb = timed(lambda: ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin))
            line = f"round {r}: attention_bwd_pv + qknorm_rope_bwd {ta:.3f} ms | attention_bwd_pv_qknorm {tb:.3f} ms (two kernels)"
            if diag:
                lib.ldmae_tune(17, 1)
                tc = timed(lambda: ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin))
                lib.ldmae_tune(17, 0)
                line += f" | {tc:.3f} ms (one pass: rowc + main + dq finish)"
            print(line)
        if diag:
            lib.ldmae_tune(17, 1)
            for dbg in (1, 2, 3):                   # timing-only ablations of the one-pass kernel (results are wrong)
                lib.ldmae_tune(18, dbg)
                td = timed(lambda: ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin))
                print(f"one pass, ablation {dbg} (1 no dQ stores / atomics, 2 no dQ product / barrier, 3 no dS image): {td:.3f} ms")
            lib.ldmae_tune(18, 0)
            # the two forms against each other and run to run
            a1 = ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin)
            a2 = ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin)
            lib.ldmae_tune(17, 0)
            b1 = ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin)
            print("one pass bitwise reproducible:", all(torch.equal(x, y) for x, y in zip(a1, a2)))
            for name, x, y in zip(("dqkv", "dwq", "dwk", "dbias"), a1, b1):
                print(f"  {name}: rel diff one pass vs two kernels {float((x.float() - y.float()).norm() / y.float().norm()):.3e}")
            print("  dq / dk / dv slots differing elements:", [float((a1[0][:, :, i] != b1[0][:, :, i]).float().mean()) for i in range(3)])
            # both against the f32 kernels (exact-f32 MFMA) on the first two images, from the same bf16 q / k / v / o / lse
            n2 = 2
            qf, kf = q2[:n2].float(), k2[:n2].float()
            vf = qkv[:n2, :, 2].float().permute(0, 2, 1, 3).contiguous()
            of, lf = ops.attention_fwd(qf, kf, vf, scale)
            dqf, dkf, dvf = ops.attention_bwd(qf, kf, vf, of, do[:n2].float(), lf, scale)
            ref = ops.qknorm_rope_bwd(dqf, dkf, dvf, qkv[:n2].float(), wq, wk, cos, sin, n2, N, H, hd)[0]
            for name, x in (("one pass", a1[0]), ("two kernels", b1[0])):
                print(f"  {name} vs the f32 kernels, per slot (dq, dk, dv):",
                      [f"{float((x[:n2, :, i].float() - ref[:, :, i]).norm() / ref[:, :, i].norm()):.3e}" for i in range(3)])
    if args.ref:
        qq, kk, vv = (x.clone().requires_grad_(True) for x in (q, k, v))
        t = timed(lambda: torch.nn.functional.scaled_dot_product_attention(qq, kk, vv))
        out = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv)
        gg = torch.randn_like(out)
        tb = timed(lambda: torch.autograd.grad(out, (qq, kk, vv), gg, retain_graph=True))
        print(f"torch SDPA: fwd {t:.3f} ms {fl / t / 1e9:7.1f} TF/s | bwd {tb:.3f} ms {2.5 * fl / tb / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
