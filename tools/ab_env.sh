#!/bin/bash
# A/B of one environment switch on the headline bench, alternating runs in one box:  tools/ab_env.sh VAR A_VALUE B_VALUE [rounds] [bench args...]
# prints ms_per_step of every run (the same process layout as the driver's bench: one python per run).
var=$1; a=$2; b=$3; rounds=${4:-3}; shift 4
for r in $(seq 1 $rounds); do
  for v in "$a" "$b"; do
    out=$(env "$var=$v" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-power "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "round $r $var=$v ms_per_step $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "loss", d.get("loss"))')"
  done
done
