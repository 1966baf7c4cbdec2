#!/usr/bin/env python3
"""Does the 256-MiB Infinity Cache keep freshly WRITTEN data for the next kernel?  Write an 800-MB f32 buffer front to back (fill), then time a read
(sum) of its LAST `w` MB against its FIRST `w` MB: if the tail of a producer's output is still cache-resident, a consumer that starts from the END of the
tensor reads it faster than one that starts from the beginning.  Also: write only `w` MB, then read the same `w` MB."""
import sys

import torch


def t_ms(fn, pre, reps=7):
    out = []
    for _ in range(reps):
        pre()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    out.sort()
    return out[len(out) // 2]


def main():
    total = 800 * 2 ** 20 // 4
    x = torch.empty(total, device="cuda")
    other = torch.empty(total, device="cuda")
    for w_mb in (32, 64, 128, 192, 256, 400):
        w = w_mb * 2 ** 20 // 4
        fill_all = lambda: x.fill_(1.0)                                   # noqa: E731
        first = t_ms(lambda: x[:w].sum(), fill_all)
        last = t_ms(lambda: x[-w:].sum(), fill_all)
        own = t_ms(lambda: x[:w].sum(), lambda: x[:w].fill_(2.0))
        cold = t_ms(lambda: x[:w].sum(), lambda: other.fill_(3.0))         # evicted by 800 MB of other writes
        gb = w_mb * 2 ** 20 / 1e9
        print(f"{w_mb:4d} MB read after an 800-MB fill: first part {first * 1e3:7.1f} us ({gb / first:6.2f} TB/s) | last part {last * 1e3:7.1f} us ({gb / last:6.2f} TB/s) | "
              f"after writing just that part {own * 1e3:7.1f} us ({gb / own:6.2f} TB/s) | cold {cold * 1e3:7.1f} us ({gb / cold:6.2f} TB/s)", flush=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def sweep():
    """A consumer the shape of rmsnorm_mod_fwd (reads f32, writes bf16) over a freshly written 805-MB tensor, in chunks: ascending (the producer's order:
    the cached tail is evicted before the scan reaches it) against descending (newest rows first)."""
    n = 256 * 1024 * 768
    x = torch.empty(n, device="cuda")
    y = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for chunks in (8, 16, 32):
        c = n // chunks
        def run(order):
            for i in order:
                y[i * c:(i + 1) * c].copy_(x[i * c:(i + 1) * c])
        asc = t_ms(lambda: run(range(chunks)), lambda: x.fill_(1.0))
        desc = t_ms(lambda: run(range(chunks - 1, -1, -1)), lambda: x.fill_(1.0))
        print(f"805 MB f32 -> bf16 in {chunks} chunks after an ascending fill: ascending {asc * 1e3:7.1f} us | descending {desc * 1e3:7.1f} us", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep()
