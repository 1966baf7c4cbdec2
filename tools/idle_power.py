#!/usr/bin/env python3
"""Package power with the GPU initialised and idle (3 s), then while a trivial kernel is relaunched back to back (clocks up, no work): the floor of the energy budget in DESIGN section 9."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.zeros(1, device="cuda")
torch.cuda.synchronize()
with bench.PowerSampler(True, 0) as ps:
    time.sleep(3.0)
print("idle, context up:", ps.summary())
x = torch.zeros(64, device="cuda")
with bench.PowerSampler(True, 0) as ps:
    t0 = time.time()
    while time.time() - t0 < 3.0:
        for _ in range(200):
            x.add_(1.0)
        torch.cuda.synchronize()
print("64-element kernel relaunched back to back:", ps.summary())
