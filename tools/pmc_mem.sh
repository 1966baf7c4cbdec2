#!/bin/bash
# usage: tools/pmc_mem.sh <kernel-substring> <one_gemm.py args...>  -- HBM-side traffic counters (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py "$@" > /dev/null 2>&1
  python3 - "$pat" <<'PY'
import csv,glob,collections,sys
f=glob.glob("/tmp/pmc/*/*counter_collection.csv")
if not f: print("no counter file"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,d in acc.items():
    for k,v in d.items(): print(f"{kn:46s} {k:16s} {sum(v)/len(v):.5g}")
PY
done
