#!/usr/bin/env python3
"""Package power and shader clock (sysfs hwmon) while one kernel family runs back to back for a few seconds.
    python tools/power_probe.py [nt|tn|attn|row]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
M = 262144
g = torch.Generator(device="cuda").manual_seed(0)
if kind == "nt":
    a = torch.randn(M, 4096, device="cuda", generator=g).to(torch.bfloat16); w = (torch.randn(768, 4096, device="cuda", generator=g) / 64).to(torch.bfloat16)
    fn = lambda: ops.gemm_nt(a, w, None)
elif kind == "tn":
    a = torch.randn(M, 2304, device="cuda", generator=g).to(torch.bfloat16); b = torch.randn(M, 768, device="cuda", generator=g).to(torch.bfloat16)
    fn = lambda: ops.gemm_tn(a, b)
elif kind == "attn":
    q, k, v = (torch.randn(256, 12, 1024, 64, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
    fn = lambda: ops.attention_fwd(q, k, v, 0.125)
else:
    x = torch.randn(M, 768, device="cuda"); y = torch.empty_like(x)
    fn = lambda: torch.add(x, 1.0, out=y)
# power / clock are read in process from the card's sysfs hwmon files (bench.PowerSampler): no rocm-smi child, so the probe is
# safe under rocprofv3 (a child would inherit the profiler's preloaded library and exec through `env` with the GPU initialised)
from bench import PowerSampler
with PowerSampler(enabled=True, device_index=0) as ps:
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
print(kind, f"{dt / n * 1e3:.3f} ms per launch")
print(ps.summary())
