#!/bin/bash
# A/B/C... of the driver's form on one box: scripts/abd.sh reps "<env A>" "<env B>" ...   (bench.py --steps 20 --warmup 5, no side passes)
reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(j['value'],1))")
    echo "$v -> $out"
  done
done
