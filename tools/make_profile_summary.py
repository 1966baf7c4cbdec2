#!/usr/bin/env python3
"""profiles/rNN_summary.md + rNN_kernel_stats.csv from a rocprofv3 --kernel-trace --stats run of bench.py.
    python tools/make_profile_summary.py <kernel_stats.csv> <steps profiled> <tag e.g. r02> "<command line>" [note]"""
import csv, shutil, sys, os
src, steps, tag, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
note = sys.argv[5] if len(sys.argv) > 5 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(src)))
shutil.copy(src, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
grp = {"bf16 NT GEMMs (gemm_nt_lines_kernel + gemm_nt_persist_kernel)": 0.0, "bf16 TN GEMM (gemm_tn_ring_kernel)": 0.0, "attention": 0.0, "row-wise HBM-bound kernels": 0.0, "other": 0.0}
for r in rows:
    n, ms = r["Name"], float(r["TotalDurationNs"]) / steps / 1e6
    if "gemm_nt_persist" in n or "gemm_nt_lines" in n: grp["bf16 NT GEMMs (gemm_nt_lines_kernel + gemm_nt_persist_kernel)"] += ms
    elif "gemm_tn_ring" in n: grp["bf16 TN GEMM (gemm_tn_ring_kernel)"] += ms
    elif "attn_" in n: grp["attention"] += ms
    elif any(k in n for k in ("rmsnorm_mod", "gate_bwd", "qknorm_rope", "swiglu_", "layernorm")): grp["row-wise HBM-bound kernels"] += ms
    else: grp["other"] += ms
with open(os.path.join(root, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write(f"# {tag} — rocprofv3 kernel summary of the headline bench\n\nCommand (on the MI355X box): `{cmd}`\n\n")
    f.write(f"{steps} profiled steps, bs=256, bf16 autocast.  GPU-busy time {tot:.1f} ms per step.  {note}\n")
    f.write(f"Full CSV: `profiles/{tag}_kernel_stats.csv`.\n\n| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:26]:
        f.write(f"| `{r['Name'][:96]}` | {int(r['Calls']) / steps:g} | {float(r['TotalDurationNs']) / steps / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    f.write("\nGroups (ms/step): " + ", ".join(f"{k} {v:.1f}" for k, v in grp.items()) + ".\n")
print(f"{tot:.2f} ms/step", grp)
