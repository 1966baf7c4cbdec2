#!/bin/bash
# Counters of the headline bench for the JSON line (verdict r04 item 7), per MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots": separate --pmc
# passes, never combined with trace domains beyond --kernel-trace:
#   pass 1  FETCH_SIZE                      (KB; doubled: gfx950 reports half of wide streaming reads)
#   pass 2  WRITE_SIZE                      (KB)
#   pass 3  SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   (matrix-pipe busy fraction and the clock the chip held)
# for the dominant MFMA kernel (the bf16 NT GEMM: gemm_nt_lines_kernel + gemm_nt_persist_kernel, all epilogues), the TN GEMM, and the dominant
# HBM-bound kernel (rmsnorm_mod_bwd_kernel<GATE>).  Writes gpurun_out/pmc_bench.json; copy it to profiles/rNN_pmc_bench.json to have bench.py
# quote it (only while the kernel sources still hash to the values recorded here).
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-power --no-extra > /tmp/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 /tmp/pmc_$tag.log; }
  echo "pass $tag done"
done
python3 - <<'PY'
import csv, glob, json, os, hashlib, collections
root = os.environ["GRAFT_REPO_ROOT"]
def key(name):
    if "gemm_nt_lines" in name or "gemm_nt_persist" in name: return "gemm_nt"
    if "gemm_tn_ring" in name: return "gemm_tn"
    if "rmsnorm_mod_bwd" in name: return "rmsnorm_mod_bwd"
    if "attn_bwd_dkdv" in name: return "attn_bwd_dkdv"
    if "attn_bwd_dq" in name: return "attn_bwd_dq"
    if "attn_fwd" in name: return "attn_fwd"
    return None
out = collections.defaultdict(dict)
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    f = glob.glob(f"/tmp/pmc_{tag}/*/*counter_collection.csv")
    if not f: print("no counter file for", tag); continue
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = key(r["Kernel_Name"])
        if k: vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    t = glob.glob(f"/tmp/pmc_{tag}/*/*kernel_trace.csv")
    if t:
        for r in csv.DictReader(open(t[0])):
            k = key(r["Kernel_Name"])
            if k: dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, d in vals.items():
        for c, v in d.items():
            out[k][c + ("_KB_avg" if c.endswith("_SIZE") else "_avg")] = sum(v) / len(v)
            out[k]["launches"] = len(v)
        if dur[k]: out[k][f"avg_duration_us_in_{tag}_pass"] = sum(dur[k]) / len(dur[k]) / 1e3
for k, d in out.items():
    if "FETCH_SIZE_KB_avg" in d or "WRITE_SIZE_KB_avg" in d:
        d["hbm_bytes_per_launch"] = (2 * d.get("FETCH_SIZE_KB_avg", 0) + d.get("WRITE_SIZE_KB_avg", 0)) * 1024
    if "GRBM_GUI_ACTIVE_avg" in d and "SQ_VALU_MFMA_BUSY_CYCLES_avg" in d:
        cyc = d["GRBM_GUI_ACTIVE_avg"] / 8                       # the counter sums the 8 XCDs
        d["kernel_cycles"] = cyc
        d["mfma_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (256 * 4 * cyc)
        us = d.get("avg_duration_us_in_SQ_VALU_MFMA_BUSY_CYCLES_pass")
        if us: d["clock_ghz"] = cyc / (us * 1e3)
def sha(files):
    h = hashlib.sha256()
    for f in files: h.update(open(os.path.join(root, "ldmae_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
res = dict(out)
res["kernel_source_sha"] = sha(("gemm.hip", "gemm_nt_lines.hip", "gemm_nt_common.h", "common.h"))      # bench.py quotes the GEMM figures only while these hash the same
res["rowwise_source_sha"] = sha(("elementwise.hip", "common.h"))
res["note"] = ("rocprofv3 --pmc over bench.py --steps 2 --warmup 1, one pass per counter set (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE); "
               "FETCH_SIZE x2 (gfx950 correction); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x GRBM_GUI_ACTIVE / 8); clock = cycles / duration of that pass")
json.dump(res, open(os.path.join(root, "gpurun_out", "pmc_bench.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
