#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels on the real LightningDiT-B/1 bs=256 shapes (random data, interleaved rounds in
one process, guide rule 24/25).  torch.matmul (hipBLASLt) is timed beside them as the known-good reference on the same
device.   python tools/bench_gemm.py [--variants 0,1,2,3,4] [--m 262144]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import _lib, ops  # noqa: E402


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
    ap.add_argument("--variants", default="0,1,2,3,4")
    ap.add_argument("--tn-variants", default="0")
    ap.add_argument("--m", type=int, default=262144)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--skip-tn", action="store_true")
    ap.add_argument("--tune5", type=int, default=0, help="start delay of every other workgroup (persistent NT kernel)")
    args = ap.parse_args()
    lib = _lib.load()
    lib.ldmae_tune(5, args.tune5)
    M = args.m
    variants = [int(v) for v in args.variants.split(",")]
    shapes = [("qkv", 2304, 768, False), ("proj", 768, 768, True), ("w12", 4096, 768, False), ("w3", 768, 2048, True),
              ("dx_w12", 768, 4096, False), ("dx_qkv", 768, 2304, False)]
    g = torch.Generator(device="cuda").manual_seed(0)
    print(f"M={M}")
    for name, N, K, gated in shapes:
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        flops = 2.0 * M * N * K
        ref = (a[:512].float() @ w.float().T + bias)
        res = {}
        xin = torch.randn(M, N, device="cuda", generator=g) if gated else None
        gate = torch.randn(M // 1024, N, device="cuda", generator=g) if gated else None
        for v in variants:
            lib.ldmae_tune(6, v)
            out = ops.gemm_nt(a, w, bias)
            err = float((out[:512].float() - ref).norm() / ref.norm())
            if gated:
                xo, y = ops.gemm_nt_gate_res(a, w, bias, xin, gate, 1024)
                refx = xin[:512] + gate[0] * ref
                err = max(err, float((xo[:512] - refx).norm() / refx.norm()))
            res[v] = [err]
        for _ in range(args.rounds):
            for v in variants:
                lib.ldmae_tune(6, v)
                res[v].append(timed(lambda: ops.gemm_nt(a, w, bias)))
                if gated:
                    res[v].append(timed(lambda: ops.gemm_nt_gate_res(a, w, bias, xin, gate, 1024)))
            res.setdefault("torch", [0.0]).append(timed(lambda: torch.matmul(a, w.T)))
        step = 2 if gated else 1
        for v, r in res.items():
            ts = r[1:]
            plain = min(ts[0::step]) if v != "torch" else min(ts)
            line = f"  NT {name:7s} N={N:5d} K={K:5d} variant={v!s:5s} err={r[0]:.1e}  bias: {plain:7.3f} ms {flops / plain / 1e9:7.1f} TF/s"
            if gated and v != "torch":
                gt = min(ts[1::2])
                line += f" | gate_res: {gt:7.3f} ms {flops / gt / 1e9:7.1f} TF/s"
            print(line)
        del a, w, xin, gate
    lib.ldmae_tune(6, 0)
    if args.skip_tn:
        return
    for name, N, K in [("dW_qkv", 2304, 768), ("dW_proj", 768, 768), ("dW_w12", 4096, 768), ("dW_w3", 768, 2048)]:
        a = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
        b = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        flops = 2.0 * M * N * K
        ref = a[:, :64].float().T @ b[:, :64].float()
        for v in [int(x) for x in args.tn_variants.split(",")]:
            lib.ldmae_tune(1, v)
            out = ops.gemm_tn(a, b)
            err = float((out[:64, :64] - ref).norm() / ref.norm())
            t = min(timed(lambda: ops.gemm_tn(a, b)) for _ in range(args.rounds))
            print(f"  TN {name:7s} N={N:5d} K={K:5d} variant={v} err={err:.1e} {t:7.3f} ms {flops / t / 1e9:7.1f} TF/s")
        t = min(timed(lambda: torch.matmul(a.T, b)) for _ in range(args.rounds))
        print(f"  TN {name:7s} torch.matmul(a.T,b) {t:7.3f} ms {flops / t / 1e9:7.1f} TF/s")
        del a, b
    lib.ldmae_tune(1, 0)


if __name__ == "__main__":
    main()
