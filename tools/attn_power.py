#!/usr/bin/env python3
"""Package power and shader clock held under SUSTAINED loops of single kernels (2.5 s each; sysfs hwmon read in process, bench.PowerSampler): is the
attention backward pair power-capped on its own, as the step is, or only inside the step?  Shapes: LightningDiT-B/1 block at bs 256, bf16, random data."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ldmae_amd import ops  # noqa: E402


def sustained(name, fn, seconds=2.5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    with bench.PowerSampler(True, 0) as ps:
        a.record()
        while time.time() - t0 < seconds:
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()
        b.record()
        torch.cuda.synchronize()
    s = ps.summary() or {}
    print(f"{name:34s} {a.elapsed_time(b) / n:8.3f} ms/call   {s.get('package_w_mean')} W mean (max {s.get('package_w_max')}, cap {s.get('package_w_cap')})   "
          f"sclk {s.get('sclk_mhz_mean')} MHz   [{s.get('samples')} samples]", flush=True)


def main():
    B, H, N, hd = 256, 12, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    scale = hd ** -0.5
    qkv = torch.randn(B, N, 3, H, hd, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B, N, H * hd, device="cuda", generator=g).to(torch.bfloat16)
    wq, wk = 1 + 0.1 * torch.randn(hd, device="cuda", generator=g), 1 + 0.1 * torch.randn(hd, device="cuda", generator=g)
    cos, sin = torch.rand(N, hd, device="cuda", generator=g), torch.rand(N, hd, device="cuda", generator=g)
    q2, k2, _ = ops.qknorm_rope_fwd(qkv, wq, wk, cos, sin, B, N, H, hd, copy_v=False)
    o2, lse2 = ops.attention_fwd_pv(q2, k2, qkv, scale)
    x = torch.randn(B * N, 768, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(2304, 768, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
    bias = torch.zeros(2304, device="cuda")
    xf = torch.randn(B * N, 768, device="cuda", generator=g)
    if len(sys.argv) > 1 and sys.argv[1] == "epilogues":
        # the NT GEMMs with fused epilogues: do they hold the cap too, or do their HBM-bound epilogue phases leave power unused?
        w12 = (torch.randn(4096, 768, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
        b12 = torch.zeros(4096, device="cuda")
        w3 = (torch.randn(768, 2048, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        w3t = w3.t().contiguous()
        wp = (torch.randn(768, 768, device="cuda", generator=g) * 0.03).to(torch.bfloat16)
        b768 = torch.zeros(768, device="cuda")
        gate = torch.randn(B, 768, device="cuda", generator=g)
        h12, hid = ops.gemm_nt_swiglu(x, w12, b12)
        xo = torch.empty_like(xf)
        for rnd in range(2):
            sustained("NT plain qkv (N 2304, K 768)", lambda: ops.gemm_nt(x, w, bias=bias, out_dtype=torch.bfloat16))
            sustained("NT SwiGLU fwd (EPI 4: N 4096, K 768)", lambda: ops.gemm_nt_swiglu(x, w12, b12))
            sustained("NT SwiGLU bwd (EPI 5: N 2048, K 768)", lambda: ops.gemm_nt_swiglu_bwd(x, w3t, h12))
            sustained("NT gated residual w3 (EPI 1: K 2048)", lambda: ops.gemm_nt_gate_res(hid, w3, b768, xf, gate, N, xout=xo))
            sustained("NT gated residual proj (EPI 1: K 768)", lambda: ops.gemm_nt_gate_res(x, wp, b768, xf, gate, N, xout=xo))
            sustained("row-wise rmsnorm_modulate_fwd", lambda: ops.rmsnorm_modulate_fwd(xf, b768 + 1, gate, gate, N, torch.bfloat16, 1e-6))
            time.sleep(1.0)
        return
    for rnd in range(2):
        sustained("attention forward", lambda: ops.attention_fwd_pv(q2, k2, qkv, scale))
        sustained("attention backward pair (fused)", lambda: ops.attention_bwd_pv_qknorm(q2, k2, qkv, o2, do, lse2, scale, wq, wk, cos, sin))
        sustained("NT GEMM qkv (M 262144, N 2304, K 768)", lambda: ops.gemm_nt(x, w, bias=bias, out_dtype=torch.bfloat16))
        sustained("f32 copy 805 MB (HBM-bound)", lambda: xf.clone())
        time.sleep(1.0)


if __name__ == "__main__":
    main()
