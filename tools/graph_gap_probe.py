#!/usr/bin/env python3
"""Do the ~6 us on either side of a persistent NT GEMM launch depend on the launch path?  20 qkv-shaped NT GEMMs
back to back: stream launches vs one hipGraph replay of the same 20 launches.    python tools/graph_gap_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldmae_amd import ops

M, D = 262144, 768
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(3 * D, D, device="cuda", generator=g) * D ** -0.5).to(torch.bfloat16)
out = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
N = 20


def burst():
    for _ in range(N):
        ops.gemm_nt(x, w, None, out=out)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / N)
    return best


t_stream = timed(burst)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    burst()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    burst()
t_graph = timed(graph.replay)
print(f"per launch: stream {t_stream * 1e3:.1f} us   graph replay {t_graph * 1e3:.1f} us   (difference {1e3 * (t_stream - t_graph):+.1f} us)")
