#!/bin/bash
# HBM traffic of the dominant kernel (gemm_nt_persist_kernel, all epilogues) during bench.py, per MI355X_MICROARCH.md "HBM":
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (KB units; FETCH_SIZE doubled: gfx950 reports half of wide streaming reads).
# Writes gpurun_out/pmc_bench.json (copy to profiles/ to have bench.py report it as roofline.traffic).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-power --no-extra > /tmp/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/*/*counter_collection.csv")
    vals = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            key = "gemm_nt" if "gemm_nt_persist" in r["Kernel_Name"] else ("gemm_tn" if "gemm_tn_ring" in r["Kernel_Name"] else None)
            if key: vals.setdefault(key, []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        out.setdefault(k, {})[c + "_KB_avg"] = sum(v) / len(v); out[k]["launches"] = len(v)
for k in out:
    out[k]["hbm_bytes_per_launch"] = (2 * out[k].get("FETCH_SIZE_KB_avg", 0) + out[k].get("WRITE_SIZE_KB_avg", 0)) * 1024
import hashlib
h = hashlib.sha256()
for f in ("gemm.hip", "gemm_nt_common.h", "common.h"):
    h.update(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "ldmae_amd", "csrc", f), "rb").read())
out["kernel_source_sha"] = h.hexdigest()[:16]      # bench.py quotes this file only while the kernel sources still hash to this
out["note"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1; FETCH_SIZE x2 (gfx950 correction)"
json.dump(out, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_bench.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
