#!/usr/bin/env python3
"""Whole-line NT GEMM kernel (gemm_nt_lines.hip, the default) against the half-line kernel (gemm_nt_persist_kernel, per-call flag
LDMAE_EPI_HALF_LINES) on the eight bf16 NT GEMMs of one LightningDiT-B/1 block at bs = 256 (M = 262144) with their real epilogues:
outputs compared BITWISE, then both timed alternately on the same box (product library, no diagnostic knobs).
    python tools/bench_lines.py [--rounds 3] [--tile]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def flat(o):
    return [t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--tile", action="store_true", help="one tile per workgroup (the data-parallel launch mode)")
    ap.add_argument("--M", type=int, default=262144)
    args = ap.parse_args()
    if args.tile:
        ops.set_gemm_launch_mode("tile")
    M, D, H = args.M, 768, 2048
    g = torch.Generator(device="cuda").manual_seed(0)
    rb = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    rf = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, dqkv, hid, h12 = rb(M, D), rb(M, 3 * D), rb(M, H), rb(M, 2 * H)
    wqkv, wproj, w12, w3 = rb(3 * D, D) * D ** -0.5, rb(D, D) * D ** -0.5, rb(2 * H, D) * D ** -0.5, rb(D, H) * H ** -0.5
    wqkv_t, w12_t, w3_t = wqkv.t().contiguous(), w12.t().contiguous(), w3.t().contiguous()
    bq, bp, b12, b3 = rf(3 * D), rf(D), rf(2 * H), rf(D)
    xin, gate = rf(M, D), rf(M // 1024, D)
    cases = [
        ("qkv      bias     N=2304 K= 768", 2.0 * M * 3 * D * D, lambda: ops.gemm_nt(x, wqkv, bq)),
        ("proj     gate_res N= 768 K= 768", 2.0 * M * D * D, lambda: ops.gemm_nt_gate_res(x, wproj, bp, xin, gate, 1024, save_y=True)),
        ("w12      swiglu   N=4096 K= 768", 2.0 * M * 2 * H * D, lambda: ops.gemm_nt_swiglu(x, w12, b12)),
        ("w3       gate_res N= 768 K=2048", 2.0 * M * D * H, lambda: ops.gemm_nt_gate_res(hid, w3, b3, xin, gate, 1024, save_y=True)),
        ("dx_w3    swiglu_b N=2048 K= 768", 2.0 * M * H * D, lambda: ops.gemm_nt_swiglu_bwd(x, w3_t, h12, with_bias=True)),
        ("dx_w12   bias     N= 768 K=4096", 2.0 * M * D * 2 * H, lambda: ops.gemm_nt(h12, w12_t, None)),
        ("dx_proj  bias     N= 768 K= 768", 2.0 * M * D * D, lambda: ops.gemm_nt(x, wproj, None)),
        ("dx_qkv   bias     N= 768 K=2304", 2.0 * M * D * 3 * D, lambda: ops.gemm_nt(dqkv, wqkv_t, None)),
    ]
    print(f"bitwise check (M = {M}, launch mode {ops.gemm_launch_mode()}):")
    ok = True
    for name, fl, fn in cases:
        ops.set_gemm_half_lines(True); ref = [t.clone() for t in flat(fn())]
        ops.set_gemm_half_lines(False); got = flat(fn())
        same = len(ref) == len(got) and all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(ref, got))
        worst = max(((a.float() - b.float()).abs().max().item() for a, b in zip(ref, got)), default=0.0)
        print(f"  {name}: {'bitwise equal' if same else f'DIFFERENT (max abs {worst:.3e})'}  ({len(got)} outputs)")
        ok &= same
        del ref, got
    res = {(c[0], v): [] for c in cases for v in (0, 1)}
    for _ in range(args.rounds):
        for name, fl, fn in cases:
            for v in (1, 0):
                ops.set_gemm_half_lines(bool(v))
                res[(name, v)].append(timed(fn))
    ops.set_gemm_half_lines(False)
    tot = {0: 0.0, 1: 0.0}
    print("timing, min of rounds.  [half] = gemm_nt_persist_kernel (64-B LDS rows), [line] = gemm_nt_lines_kernel (128-B LDS rows):")
    for name, fl, fn in cases:
        th, tl = min(res[(name, 1)]), min(res[(name, 0)]); tot[1] += th; tot[0] += tl
        print(f"{name}:  [half] {th:6.3f} ms {fl / th / 1e9:7.1f} TF/s   [line] {tl:6.3f} ms {fl / tl / 1e9:7.1f} TF/s   {100 * (tl / th - 1):+5.1f} %")
    print(f"block total:  [half] {tot[1]:6.3f} ms   [line] {tot[0]:6.3f} ms   {100 * (tot[0] / tot[1] - 1):+5.1f} %")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
