#!/usr/bin/env python3
"""The four bf16 TN (weight-gradient) GEMMs of one LightningDiT-B/1 block at bs = 256 (M = 262144 token rows), split-K reduce included.
A/B of two BUILDS: run twice in one gpurun call with LDMAE_HIP_LIB naming the library.
    python tools/bench_tn.py [--rounds 3]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    M, D, H = 262144, 768, 2048
    g = torch.Generator(device="cuda").manual_seed(0)
    rb = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
    x, dqkv, hid, dh12, dy = rb(M, D), rb(M, 3 * D), rb(M, H), rb(M, 2 * H), rb(M, D)
    outs = {k: torch.zeros(*s, device="cuda") for k, s in dict(qkv=(3 * D, D), proj=(D, D), w12=(2 * H, D), w3=(D, H)).items()}
    cases = [
        ("dW qkv   N=2304 K= 768", 2.0 * M * 3 * D * D, lambda: ops.gemm_tn(dqkv, x, out=outs["qkv"], beta=1.0)),
        ("dW proj  N= 768 K= 768", 2.0 * M * D * D, lambda: ops.gemm_tn(dy, x, out=outs["proj"], beta=1.0)),
        ("dW w12   N=4096 K= 768", 2.0 * M * 2 * H * D, lambda: ops.gemm_tn(dh12, x, out=outs["w12"], beta=1.0)),
        ("dW w3    N= 768 K=2048", 2.0 * M * D * H, lambda: ops.gemm_tn(dy, hid, out=outs["w3"], beta=1.0)),
    ]
    res = {c[0]: [] for c in cases}
    for _ in range(args.rounds):
        for name, fl, fn in cases:
            res[name].append(timed(fn))
    tot = 0.0
    print(f"library: {os.environ.get('LDMAE_HIP_LIB', 'ldmae_amd/libldmae_hip.so')}")
    for name, fl, fn in cases:
        t = min(res[name]); tot += t
        print(f"{name}: {t:6.3f} ms {fl / t / 1e9:7.1f} TF/s")
    print(f"block total: {tot:6.3f} ms   checksum {sum(float(o.double().abs().sum()) for o in outs.values()):.6e}")


if __name__ == "__main__":
    main()
