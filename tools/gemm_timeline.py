#!/usr/bin/env python3
"""Timeline of the persistent NT GEMM: per workgroup and tile, s_memrealtime (100 MHz) stamps at tile start, main-loop end,
epilogue issued, stores drained.   python tools/gemm_timeline.py N K [delay]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))   # A/B knobs live in the diagnostic build only (make -C ldmae_amd/csrc diag)
from ldmae_amd import _lib, ops
N, K = int(sys.argv[1]), int(sys.argv[2])
delay = int(sys.argv[3]) if len(sys.argv) > 3 else 0
EPI = os.environ.get("EPI", "bias")       # bias | swiglu | swiglu_bwd | gate
M = 262144
lib = _lib.load()
lib.ldmae_tune(8, 1); lib.ldmae_tune(5, delay)
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
ntiles = (M // 256) * (N // 256)
iters = (ntiles + 255) // 256
if EPI == "swiglu":
    run = lambda: ops.gemm_nt_swiglu(a, w, bias)
elif EPI == "swiglu_bwd":
    h12 = torch.randn(M, 2 * N, device="cuda", generator=g).to(torch.bfloat16)
    run = lambda: ops.gemm_nt_swiglu_bwd(a, w, h12)
elif EPI == "gate":
    xin = torch.randn(M, N, device="cuda", generator=g)
    gate = torch.randn(M // 1024, N, device="cuda", generator=g)
    run = lambda: ops.gemm_nt_gate_res(a, w, bias, xin, gate, 1024)
else:
    run = lambda: ops.gemm_nt(a, w, bias)
for _ in range(3): run()
buf = torch.zeros(iters * 256 * 2 * 4, dtype=torch.int64, device="cuda")
lib.ldmae_debug_nt_stamps(buf.data_ptr())
run()
torch.cuda.synchronize()
lib.ldmae_debug_nt_stamps(None)
s = buf.cpu().numpy().reshape(iters, 256, 2, 4).astype(np.float64)
full = iters - 1 if ntiles % 256 else iters
s = s[:full] / 100.0                     # us
t0 = s[0, :, :, 0].min()
s -= t0
tile = s[1:, :, 0, 0] - s[:-1, :, 0, 0]
main = s[:, :, :, 1] - s[:, :, :, 0]
epi = s[:, :, :, 2] - s[:, :, :, 1]
drain = s[:, :, :, 3] - s[:, :, :, 2]
print(f"EPI={EPI} N={N} K={K} delay={delay}: {full} full rounds, kernel span {s[..., 3].max():.1f} us")
print(f"tile period  mean {tile.mean():6.2f} us  p10 {np.percentile(tile,10):6.2f}  p90 {np.percentile(tile,90):6.2f}")
for nm, x in (("main loop", main), ("epilogue", epi), ("drain", drain)):
    print(f"{nm:12s} grpA mean {x[:, :, 0].mean():6.2f} us  grpB mean {x[:, :, 1].mean():6.2f} us   p10 {np.percentile(x,10):6.2f} p90 {np.percentile(x,90):6.2f}")
# phase spread: when do workgroups start their epilogue within a round?
for r in (1, full // 2, full - 1):
    e = s[r, :, 0, 1]
    print(f"round {r}: epilogue start min {e.min():7.1f} max {e.max():7.1f} std {e.std():5.2f} us; start of tile spread {s[r,:,0,0].std():5.2f}")
# concurrency: how many workgroups are inside [main-loop end, stores drained] at a time
ev = np.concatenate([np.stack([s[:, :, 0, 1].ravel(), np.ones(full * 256)], 1), np.stack([s[:, :, 0, 3].ravel(), -np.ones(full * 256)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0], append=ev[-1, 0])
print(f"time-weighted mean # of workgroups in epilogue: {np.sum(conc * dt) / dt.sum():.1f}; while any: {np.sum(conc * dt) / dt[conc > 0].sum():.1f}")
if os.environ.get("TL"):
    # per-K-step phase stamps (shader clock) of the second tile: diagnostic build (tune key 7)
    nk = K // 32
    buf = torch.zeros(iters * 256 * 2 * 4 + 256 * 1024 + 256 * 256, dtype=torch.int64, device="cuda")
    lib.ldmae_tune(7, 1)
    lib.ldmae_debug_nt_stamps(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.ldmae_debug_nt_stamps(None)
    lib.ldmae_tune(7, 0)
    tl = buf.cpu().numpy()[256 * 1024:256 * 1024 + 256 * 256].reshape(256, 2, 32, 4)[:, :, :min(nk, 32)].astype(np.float64)
    L = tl[..., 1] - tl[..., 0]            # issue loads + fragment reads landed
    B1 = tl[..., 2] - tl[..., 1]           # (grpB: wait for next stage) + barrier
    Mm = tl[..., 3] - tl[..., 2]           # MFMA phase
    B2 = tl[:, :, 1:, 0] - tl[:, :, :-1, 3]  # (grpA: wait for next stage) + barrier
    step = tl[:, :, 1:, 0] - tl[:, :, :-1, 0]
    for gname, gi in (("grpA", 0), ("grpB", 1)):
        print(f"{gname}: K-step {step[:, gi].mean():7.0f} cyc | load+frag {L[:, gi].mean():6.0f}  wait/barrier {B1[:, gi].mean():6.0f}  mfma {Mm[:, gi].mean():6.0f}  wait/barrier {B2[:, gi].mean():6.0f}")
    print("per-K-step (grpA, workgroup 5):", " ".join(f"{x:.0f}" for x in step[5, 0]))
