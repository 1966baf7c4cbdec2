#!/usr/bin/env python3
"""Randomised sweep of the whole-line NT GEMM (gemm_nt_lines_kernel, the default) against the half-line kernel it replaced
(LDMAE_EPI_HALF_LINES): bitwise-equal outputs for every fused epilogue and both launch modes on random shapes (M, N multiples of 8, K of 64),
plus the arithmetic against an f32 matmul.  The fixed-shape version is tests/test_gpu_kernels.py; this is the wide net.
    python tools/fuzz_lines.py [shapes=40] [seed=0]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def main_f16(n, rnd):
    """fp16 instantiation (no half-line twin): every forward epilogue against f32 torch arithmetic on the same fp16 operands."""
    F16 = torch.float16
    g = torch.Generator(device="cuda").manual_seed(6)
    bad = 0
    rel = lambda x, y: float((x.float() - y).norm() / y.norm())      # noqa: E731
    for i in range(n):
        M = 8 * rnd.choice([1, 2, 31, 33, 64, 257, 513, 2049, rnd.randint(1, 4000)])
        N = 8 * rnd.choice([1, 3, 24, 32, 33, 72, 96, 288, rnd.randint(1, 400)])
        K = 64 * rnd.choice([1, 2, 3, 4, 12, 18, rnd.randint(1, 40)])
        a = torch.randn(M, K, device="cuda", generator=g).to(F16)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(F16)
        bias, xin = torch.randn(N, device="cuda", generator=g), torch.randn(M, N, device="cuda", generator=g)
        y = a.float() @ w.float().T + bias
        errs = {"bias": rel(ops.gemm_nt(a, w, bias), y), "bias_f32out": rel(ops.gemm_nt(a, w, bias, out_dtype=F32), y),
                "gate_res": rel(ops.gemm_nt_gate_res(a, w, bias, xin, None, M, save_y=False, y_dtype=F32)[0], xin + y),
                "gelu": rel(ops.gemm_nt_gelu(a, w, bias, save_pre=False)[0], torch.nn.functional.gelu(y))}
        ok = all(e < 1.5e-3 for e in errs.values())
        bad += not ok
        print(f"{i:3d}  M {M:6d} N {N:5d} K {K:5d}  {'ok' if ok else 'FAIL'}  " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()), flush=True)
    print(f"{n - bad} / {n} fp16 shapes within 1.5e-3 of the f32 arithmetic on every epilogue")
    sys.exit(1 if bad else 0)


def main_qkv(n, rnd):
    """ldmae_gemm_nt_qkv_rope (QK-norm / RoPE in the qkv GEMM's epilogue) against ldmae_gemm_nt + ldmae_qknorm_rope_fwd on random covered shapes: bitwise."""
    g = torch.Generator(device="cuda").manual_seed(7)
    bad = 0
    for i in range(n):
        N = 128 * rnd.choice([1, 2, 3, 5, 8, rnd.randint(1, 12)])
        B = rnd.choice([1, 2, 3, 4, 7, rnd.randint(1, 24)])
        if (B * N) % 256:
            B *= 2
        H = 4 * rnd.choice([1, 2, 3, 4, 6])
        K = 64 * rnd.choice([1, 2, 3, 12, 18, rnd.randint(1, 24)])
        norm, with_bias, raw, tile = rnd.random() < 0.6, rnd.random() < 0.7, rnd.random() < 0.7, rnd.random() < 0.4
        a = torch.randn(B * N, K, device="cuda", generator=g).to(BF16)
        w = (torch.randn(3 * H * 64, K, device="cuda", generator=g) * K ** -0.5).to(BF16)
        bias = torch.randn(3 * H * 64, device="cuda", generator=g) * 0.2 if with_bias else None
        wq = 1 + 0.3 * torch.randn(64, device="cuda", generator=g) if norm else None
        wk = 1 + 0.3 * torch.randn(64, device="cuda", generator=g) if norm else None
        cos, sin = torch.randn(N, 64, device="cuda", generator=g), torch.randn(N, 64, device="cuda", generator=g)      # arbitrary tables: no pair structure assumed
        assert ops.gemm_nt_qkv_rope_ok(a, w, B, N, H, 64)
        ops.set_gemm_launch_mode("tile" if tile else "persistent")
        ref = ops.gemm_nt(a, w, bias)
        qr, kr, _ = ops.qknorm_rope_fwd(ref, wq, wk, cos, sin, B, N, H, 64, 1e-6, copy_v=False)
        qkv, q2, k2 = ops.gemm_nt_qkv_rope(a, w, bias, wq, wk, cos, sin, B, N, H, 64, 1e-6, store_raw_qk=raw)
        ops.set_gemm_launch_mode("auto")
        ok = torch.equal(q2, qr) and torch.equal(k2, kr)
        ok = ok and (torch.equal(qkv, ref) if raw else torch.equal(qkv.view(B * N, 3, H * 64)[:, 2], ref.view(B * N, 3, H * 64)[:, 2]))
        bad += not ok
        print(f"{i:3d}  B {B:3d} N {N:5d} H {H:3d} K {K:5d}  norm {int(norm)} bias {int(with_bias)} raw {int(raw)} tile {int(tile)}  {'ok' if ok else 'FAIL'}", flush=True)
    print(f"{n - bad} / {n} shapes bitwise equal to the GEMM + qknorm_rope_fwd pair")
    sys.exit(1 if bad else 0)


def main():
    if "--qkv" in sys.argv:
        sys.argv.remove("--qkv")
        return main_qkv(int(sys.argv[1]) if len(sys.argv) > 1 else 40, random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    if "--f16" in sys.argv:
        sys.argv.remove("--f16")
        return main_f16(int(sys.argv[1]) if len(sys.argv) > 1 else 40, random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    g = torch.Generator(device="cuda").manual_seed(5)
    rb = lambda *s: torch.randn(*s, device="cuda", generator=g).to(BF16)      # noqa: E731
    rf = lambda *s: torch.randn(*s, device="cuda", generator=g)               # noqa: E731
    flat = lambda o: [t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)]      # noqa: E731
    bad = 0
    for i in range(n):
        M = 8 * rnd.choice([1, 2, 3, 31, 32, 33, 64, 100, 257, 512, 1000, 2049, 4096, rnd.randint(1, 6000)])
        N = 8 * rnd.choice([1, 2, 24, 31, 32, 33, 64, 72, 96, 144, 288, 512, rnd.randint(1, 600)])
        K = 64 * rnd.choice([1, 1, 2, 2, 3, 4, 5, 12, 18, 32, 36, 64, rnd.randint(1, 48)])
        a, w, bias = rb(M, K), rb(N, K) * K ** -0.5, rf(N)
        T = 8
        xin, gate, pos, pre, h12 = rf(M, N), rf(M // T, N), rf(T, N), rb(M, N), rb(M, 2 * N)
        cases = {
            "bias": lambda: ops.gemm_nt(a, w, bias), "bias_f32out": lambda: ops.gemm_nt(a, w, bias, out_dtype=F32),
            "gate_res": lambda: ops.gemm_nt_gate_res(a, w, bias, xin, gate, T), "gelu": lambda: ops.gemm_nt_gelu(a, w, bias, save_pre=True),
            "pos": lambda: ops.gemm_nt_pos(a, w, bias, pos, T), "gelu_bwd": lambda: ops.gemm_nt_gelu_bwd(a, w, pre),
            "swiglu_bwd": lambda: ops.gemm_nt_swiglu_bwd(a, w, h12, with_bias=True),
        }
        if N % 256 == 0 and N >= 512:
            cases["swiglu"] = lambda: ops.gemm_nt_swiglu(a, w, bias)
        fails = []
        try:
            for mode in ("persistent", "tile"):
                ops.set_gemm_launch_mode(mode)
                for name, fn in cases.items():
                    ops.set_gemm_half_lines(True)
                    ref = [t.clone() for t in flat(fn())]
                    ops.set_gemm_half_lines(False)
                    got = flat(fn())
                    if not (len(ref) == len(got) and all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(ref, got))):
                        fails.append((name, mode))
            ops.set_gemm_half_lines(False)
            want = a.float() @ w.float().T + bias
            err = float((ops.gemm_nt(a, w, bias).float() - want).norm() / want.norm())
        finally:
            ops.set_gemm_half_lines(False)
            ops.set_gemm_launch_mode("persistent")
        ok = not fails and err < 1e-2
        bad += not ok
        print(f"{i:3d}  M {M:6d} N {N:5d} K {K:5d}  {'ok' if ok else 'FAIL ' + str(fails)}  rel err vs f32 {err:.2e}", flush=True)
    print(f"{n - bad} / {n} shapes bitwise equal on all epilogues and launch modes")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
