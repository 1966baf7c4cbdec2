#!/usr/bin/env python3
"""Can a weight-gradient (TN) GEMM on a side stream run UNDER an HBM-bound row kernel on the main stream?  (round-2 verdict, item 3a)
Times, at the DiT-B/1 bs=256 shapes (M = 262144 token rows): each kernel alone, the pair back to back on one stream, and the pair on two
streams, for the product TN kernel (8 waves x 231 VGPRs + 160 KiB LDS: owns the whole CU) and for the small-footprint fallback TN kernel
(4 waves x 154 VGPRs, 64 KiB LDS: leaves room for other workgroups on the CU).
    python tools/overlap_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))
from ldmae_amd import _lib, ops

lib = _lib.load()
M, D, N3, T, B = 262144, 768, 2304, 1024, 256
g = torch.Generator(device="cuda").manual_seed(0)
bf = torch.bfloat16
dqkv = torch.randn(M, N3, device="cuda", generator=g).to(bf)
xm1 = torch.randn(M, D, device="cuda", generator=g).to(bf)
dout = torch.randn(M, D, device="cuda", generator=g).to(bf)
x = torch.randn(M, D, device="cuda", generator=g)
dx = torch.randn(M, D, device="cuda", generator=g)
y = torch.randn(M, D, device="cuda", generator=g).to(bf)
w = torch.randn(D, device="cuda", generator=g)
mod = torch.randn(B, 6 * D, device="cuda", generator=g) * 0.1
dmod = torch.empty_like(mod)
rstd = torch.rand(M, device="cuda", generator=g) + 0.5
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def row():
    ops.rmsnorm_modulate_bwd_gate(dout, x, w, mod[:, D:2 * D], rstd, dx, dmod[:, 0:D], dmod[:, D:2 * D], y, mod[:, 2 * D:3 * D], dmod[:, 2 * D:3 * D], T, bf)


def tn(slot="tn"):
    ops.gemm_tn(dqkv, xm1, ws_slot=slot)


def nt():
    ops.gemm_nt(dqkv, wT)


wT = (torch.randn(D, N3, device="cuda", generator=g) / 48).to(bf)      # dxm1 = dqkv . Wqkv^T^T: the NT GEMM that runs next on the critical path


def wall(fn, it=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def two_streams(other):
    def f():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            tn("tn_side")
        other()
        main.wait_stream(side)
    return f


def serial(other):
    def f():
        tn(); other()
    return f


for variant, name in ((0, "ring TN (product: 8 waves x 231 VGPR, 160 KiB LDS)"), (1, "fallback TN (4 waves x 154 VGPR, 64 KiB LDS)")):
    lib.ldmae_tune(1, variant)
    t_tn, t_row, t_nt = wall(tn), wall(row), wall(nt)
    print(f"{name}: tn {t_tn:.3f} ms  row kernel {t_row:.3f} ms  nt gemm {t_nt:.3f} ms")
    for oname, other, t_o in (("row kernel", row, t_row), ("nt gemm", nt, t_nt)):
        t_ser, t_par = wall(serial(other)), wall(two_streams(other))
        print(f"    tn + {oname}: one stream {t_ser:.3f} ms  two streams {t_par:.3f} ms  (hidden {t_ser - t_par:+.3f} ms of the "
              f"{min(t_tn, t_o):.3f} ms that could overlap)")
lib.ldmae_tune(1, 0)
