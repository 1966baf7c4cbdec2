#!/bin/bash
# usage: tools/pmc.sh <kernel-substring> <one_gemm.py args...>   -- collects a few SQ counters for matching kernels
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  rm -rf /tmp/pmc
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py "$@" > /dev/null 2>&1
  python3 - "$pat" <<'PY'
import csv,glob,collections,sys
f=glob.glob("/tmp/pmc/*/*counter_collection.csv")
if not f: print("no counter file"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,d in acc.items():
    for k,v in d.items(): print(f"{kn:50s} {k:28s} {sum(v)/len(v):.4g}")
PY
done
