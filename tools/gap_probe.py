#!/usr/bin/env python3
"""Launch-gap probe: [small kernel, NT GEMM slot, small kernel] x 20 for tune key 11 = 0 (the real persistent NT GEMM),
1 (an EMPTY kernel with the same launch configuration and arguments) and 2 (the real kernel with ntiles = 0: every workgroup returns at once).
Run under rocprofv3 --kernel-trace and read the gaps around the NT slot with tools/gap_probe.py --analyse <kernel_trace.csv>.
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/gap_probe.py"""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    import csv, collections
    rows = list(csv.DictReader(open(sys.argv[2]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    g = collections.defaultdict(list)
    for a, b, c in zip(rows, rows[1:], rows[2:]):
        if "cast_kernel" in a["Kernel_Name"] and "cast_kernel" in c["Kernel_Name"] and ("gemm_nt" in b["Kernel_Name"]):
            g[b["Kernel_Name"][:44] + (" (ntiles=0)" if int(b["End_Timestamp"]) - int(b["Start_Timestamp"]) < 20000 and "persist" in b["Kernel_Name"] else "")].append(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, (int(c["Start_Timestamp"]) - int(b["End_Timestamp"])) / 1e3,
                                             (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3))
    for k, v in g.items():
        n = len(v)
        print(f"{k:58s} n={n:3d}  gap before {sum(x[0] for x in v) / n:6.2f} us  after {sum(x[1] for x in v) / n:6.2f} us  kernel {sum(x[2] for x in v) / n:8.1f} us")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LDMAE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ldmae_amd", "libldmae_hip_diag.so"))
from ldmae_amd import _lib, ops
lib = _lib.load()
M, D = 262144, 768
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(D, D, device="cuda", generator=g) * D ** -0.5).to(torch.bfloat16)
out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
small = torch.randn(1 << 18, device="cuda")
sb = torch.empty(1 << 18, device="cuda", dtype=torch.bfloat16)


def burst():
    for _ in range(20):
        lib.ldmae_cast(_lib.F32, _lib.BF16, small.data_ptr(), sb.data_ptr(), small.numel(), _lib.stream())
        ops.gemm_nt(x, w, None, out=out)
        lib.ldmae_cast(_lib.F32, _lib.BF16, small.data_ptr(), sb.data_ptr(), small.numel(), _lib.stream())


# replayed as ONE hipGraph per mode, so that the host's launch rate plays no part in the gaps
for mode in (0, 1, 2):
    lib.ldmae_tune(11, mode)
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        burst()
    torch.cuda.current_stream().wait_stream(s_)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        burst()
    gr.replay(); gr.replay()
    torch.cuda.synchronize()
lib.ldmae_tune(11, 0)
