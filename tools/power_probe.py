#!/usr/bin/env python3
"""Package power and shader clock (rocm-smi) while one kernel family runs back to back for a few seconds.
    python tools/power_probe.py [nt|tn|attn|row]"""
import os, subprocess, sys, threading, time
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
samples, stop = [], False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip().splitlines()[-1] if out.strip() else "(empty)")
        except Exception as e:
            samples.append(f"err {e}")
        time.sleep(0.3)
th = threading.Thread(target=poll); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 4.0:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
stop = True; th.join()
print(kind, f"{dt / n * 1e3:.3f} ms per launch")
hdr = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print(hdr[0] if hdr else "(no header)")
for s in samples[2:8]: print(s)
