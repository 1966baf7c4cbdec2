#!/usr/bin/env python3
"""Deferred-epilogue NT GEMM (csrc/probe/gemm_nt_defer.hip) against gemm_nt_persist_kernel with the fused epilogue (diag tune key 12: 1 = deferred kernel, 0 = fused):
bitwise equality of every output, several shapes incl. the bench ones, repeated launches (race screen).
    LDMAE_HIP_LIB=.../libldmae_hip_diag.so python tools/test_defer.py [--big]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))
from ldmae_amd import _lib, ops

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
rb = lambda *s: torch.randn(*s, device="cuda", generator=g).to(torch.bfloat16)
rf = lambda *s: torch.randn(*s, device="cuda", generator=g)
big = "--big" in sys.argv
ok = True
shapes = [(256 * 96, 768, 768, 1024), (256 * 40, 768, 2048, 256), (256 * 24, 1152, 1152, 512)]
if big:
    shapes += [(262144, 768, 768, 1024), (262144, 768, 2048, 1024)]
for M, N, K, T in shapes:
    a, w, bias = rb(M, K), rb(N, K) * K ** -0.5, rf(N)
    xin, gate = rf(M, N), rf(M // T, N)
    outs = {}
    for v in (0, 1):
        lib.ldmae_tune(12, v)
        res = []
        for rep in range(3 if v == 1 else 1):
            xo, y = ops.gemm_nt_gate_res(a, w, bias, xin, gate, T)
            res.append((xo, y))
        outs[v] = res
    lib.ldmae_tune(12, 0)
    ref_xo, ref_y = outs[0][0]
    for i, (xo, y) in enumerate(outs[1]):
        e1, e2 = torch.equal(xo, ref_xo), torch.equal(y, ref_y)
        ok &= e1 and e2
        print(f"gate_res M={M} N={N} K={K} rep{i}: xout {'==' if e1 else '!='}  y {'==' if e2 else '!='}" +
              ("" if e1 else f"  nbad={int((xo != ref_xo).sum())} maxdiff={float((xo - ref_xo).abs().max()):.3e}"))
    # in-place residual
    lib.ldmae_tune(12, 1)
    xi2 = xin.clone()
    xo2, _ = ops.gemm_nt_gate_res(a, w, bias, xi2, gate, T, xout=xi2)
    e = torch.equal(xo2, ref_xo); ok &= e
    print(f"   in place: {'==' if e else '!='}")
    lib.ldmae_tune(12, 0)
sh2 = [(256 * 96, 2048, 768), (256 * 40, 1024, 1152)]
if big:
    sh2 += [(262144, 2048, 768)]
for M, Hs, K in sh2:
    a, w12, b12 = rb(M, K), rb(2 * Hs, K) * K ** -0.5, rf(2 * Hs)
    outs = {}
    for v in (0, 1):
        lib.ldmae_tune(12, v)
        outs[v] = [ops.gemm_nt_swiglu(a, w12, b12) for _ in range(3 if v == 1 else 1)]
    lib.ldmae_tune(12, 0)
    rh12, rhid = outs[0][0]
    for i, (h12, hid) in enumerate(outs[1]):
        e1, e2 = torch.equal(h12, rh12), torch.equal(hid, rhid)
        ok &= e1 and e2
        print(f"swiglu M={M} Hs={Hs} K={K} rep{i}: h12 {'==' if e1 else '!='}  hid {'==' if e2 else '!='}" +
              ("" if e2 else f"  nbad={int((hid != rhid).sum())}"))
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
