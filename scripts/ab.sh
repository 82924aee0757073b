#!/bin/bash
# A/B on one box: scripts/ab.sh "<env A>" "<env B>" [reps]   (bench.py --steps 300, no side passes)
reps=${3:-3}
for r in $(seq $reps); do
  for v in "$1" "$2"; do
    out=$(env $v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-config-legs --no-session-leg --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(j['value'],1), j['host_blocked_ms_per_step'])")
    echo "$v -> $out"
  done
done
