#!/bin/bash
# usage: tools/_ab.sh libA libB  -> per-kernel ms/step of bench.py under rocprofv3 for two builds on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for l in "$@"; do
  rm -rf gpurun_out/ab_$l
  LDMAE_HIP_LIB=$GRAFT_REPO_ROOT/ldmae_amd/libldmae_hip$l.so rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$l -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-power --no-extra > gpurun_out/ab_$l.log 2>&1
done
python3 - "$@" <<'PY'
import csv,glob,sys
def load(tag):
    f=glob.glob(f'gpurun_out/ab_{tag}/**/*kernel_stats.csv',recursive=True)[0]
    return {r['Name'][:64]:(float(r['TotalDurationNs'])/4e6,float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f))}
a,b=load(sys.argv[1]),load(sys.argv[2])
print('total ms/step', round(sum(v[0] for v in a.values()),2), round(sum(v[0] for v in b.values()),2))
for k in sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[0]+b.get(k,(0,0))[0]))[:22]:
    x=a.get(k,(0,0)); y=b.get(k,(0,0))
    print(f"{k:64s} {x[0]:7.2f} | {y[0]:7.2f} ms/step   avg {x[1]:8.1f} | {y[1]:8.1f} us")
PY
