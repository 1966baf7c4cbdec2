#!/bin/bash
# usage: tools/_ab_env.sh "LDMAE_TUNE=10=64" "LDMAE_TUNE=10=16" ... -> per-kernel ms/step of bench.py under rocprofv3 for each env setting (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# LDMAE_TUNE knobs exist in the diagnostic build only
export LDMAE_HIP_LIB=${LDMAE_HIP_LIB:-$GRAFT_REPO_ROOT/ldmae_amd/libldmae_hip_diag.so}
i=0
for e in "$@"; do
  i=$((i+1)); rm -rf gpurun_out/abe_$i
  export $e
  rocprofv3 --kernel-trace --stats -d gpurun_out/abe_$i -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-power --no-extra > gpurun_out/abe_$i.log 2>&1
done
python3 - $i <<'PY'
import csv,glob,sys
n=int(sys.argv[1])
def load(i):
    f=glob.glob(f'gpurun_out/abe_{i}/**/*kernel_stats.csv',recursive=True)[0]
    return {r['Name'][:56]:float(r['TotalDurationNs'])/4e6 for r in csv.DictReader(open(f))}
d=[load(i) for i in range(1,n+1)]
print('total ms/step', [round(sum(x.values()),2) for x in d])
keys=sorted(set().union(*d), key=lambda k:-sum(x.get(k,0) for x in d))[:18]
for k in keys: print(f"{k:56s}", "  ".join(f"{x.get(k,0):7.2f}" for x in d))
PY
