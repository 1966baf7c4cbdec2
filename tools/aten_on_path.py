#!/usr/bin/env python3
"""Which ATen kernels still run inside the headline train step, from where: torch.profiler over two steps of bench.py's own step function
(bs 256, bf16), grouped by operator + input shapes + the innermost repo frame.  `python tools/aten_on_path.py [batch]`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    model, opt, reducer, transport = bench.build(dev, B)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, 16, 32, 32, device=dev, generator=g)
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)
    for _ in range(3):
        bench.train_step(model, opt, reducer, transport, x, y)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(2):
            bench.train_step(model, opt, reducer, transport, x, y)
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if not e.name.startswith("aten::") or e.device_time_total <= 0 or e.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
            continue
        frame = next((f for f in (e.stack or []) if "/ldmae_amd/" in f or "bench.py" in f), (e.stack or ["?"])[0] if e.stack else "?")
        key = (e.name, str(e.input_shapes), frame.replace(ROOT + "/", ""))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += e.device_time_total
    tot = 0.0
    for (name, shapes, frame), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{n / 2:6.1f}/step {us / 2:9.1f} us/step  {name:28s} {shapes[:90]:90s} {frame[:110]}")
        tot += us / 2
    print(f"ATen device time per step: {tot:.1f} us")


if __name__ == "__main__":
    main()
