#!/bin/bash
# A/B/C... of the config legs on one box: scripts/ablegs.sh reps "<env A>" "<env B>" ...   (bench.py default steps, config legs only)
reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(j['value'],1), {k:round(v['value'],1) for k,v in j.get('configs',{}).items()})")
    echo "$v -> $out"
  done
done
