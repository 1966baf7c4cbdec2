#!/bin/bash
# usage: tools/pmc_set.sh <kernel-substring> "<counter set 1>" ["<counter set 2>" ...] -- <one_gemm.py args...>
# one rocprofv3 --pmc pass per set over tools/one_gemm.py; prints the per-launch mean of every counter for matching kernels
pat=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
for set in "${sets[@]}"; do
  rm -rf /tmp/pmc
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py "$@" > /tmp/pmc.log 2>&1
  python3 - "$pat" <<'PY'
import csv,glob,collections,sys
f=glob.glob("/tmp/pmc/*/*counter_collection.csv")
if not f: print("no counter file"); print(open("/tmp/pmc.log").read()[-800:]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,d in acc.items():
    for k,v in d.items(): print(f"{kn:50s} {k:32s} {sum(v)/len(v):.6g}")
t=glob.glob("/tmp/pmc/*/*kernel_trace.csv")
if t:
    dur=collections.defaultdict(list)
    for r in csv.DictReader(open(t[0])):
        if sys.argv[1] in r["Kernel_Name"]: dur[r["Kernel_Name"][:48]].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
    for kn,v in dur.items(): print(f"{kn:50s} {'duration_ns (this pass)':32s} {sum(v)/len(v):.6g}")
PY
done
