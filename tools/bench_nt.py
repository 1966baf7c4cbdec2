#!/usr/bin/env python3
"""The eight bf16 NT GEMMs of one LightningDiT-B/1 block (bs=256: M = 262144 token rows) with their real epilogues, timed
for each value of a tune key (default key 8: 1 = persistent workgroups, 2 = one tile per workgroup).
    python tools/bench_nt.py [--key 8] [--values 1,2] [--rounds 3]"""
import os, argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--key=-1" not in sys.argv:          # A/B knobs live in the diagnostic build only (make -C ldmae_amd/csrc diag); --key=-1: no knob, time
    os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))   # whatever library LDMAE_HIP_LIB names (A/B of two BUILDS: run twice in one gpurun call)
from ldmae_amd import _lib, ops


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", type=int, default=8)
    ap.add_argument("--values", default="1,2")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    lib = _lib.load()
    vals = [int(v) for v in args.values.split(",")] if args.key >= 0 else [0]
    M, D, H = 262144, 768, 2048
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
        ("dx_w3    swiglu_b N=2048 K= 768", 2.0 * M * H * D, lambda: ops.gemm_nt_swiglu_bwd(x, w3_t, h12)),
        ("dx_w12   bias     N= 768 K=4096", 2.0 * M * D * 2 * H, lambda: ops.gemm_nt(h12, w12_t, None)),
        ("dx_proj  bias     N= 768 K= 768", 2.0 * M * D * D, lambda: ops.gemm_nt(x, wproj, None)),
        ("dx_qkv   bias     N= 768 K=2304", 2.0 * M * D * 3 * D, lambda: ops.gemm_nt(dqkv, wqkv_t, None)),
    ]
    res = {(c[0], v): [] for c in cases for v in vals}
    for _ in range(args.rounds):
        for name, fl, fn in cases:
            for v in vals:
                if args.key >= 0:
                    lib.ldmae_tune(args.key, v)
                res[(name, v)].append(timed(fn))
    if args.key >= 0:
        lib.ldmae_tune(args.key, 0)
    tot = {v: 0.0 for v in vals}
    for name, fl, fn in cases:
        line = f"{name}:"
        for v in vals:
            t = min(res[(name, v)]); tot[v] += t
            line += f"  [{v}] {t:6.3f} ms {fl / t / 1e9:7.1f} TF/s"
        print(line)
    print("block total:" + "".join(f"  [{v}] {tot[v]:6.3f} ms" for v in vals))


if __name__ == "__main__":
    main()
