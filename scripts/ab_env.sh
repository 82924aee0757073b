#!/bin/bash
# A/B of environment settings on one box, both forms: scripts/ab_env.sh reps "ENV=a" "ENV=b" ...  -> driver form (20 steps after 5) and 300 steps
reps=$1; shift
summ='import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(j["value"],1), j["config"]["surfels_per_map"])'
for r in $(seq $reps); do
  for v in "$@"; do
    d=$(env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "$summ")
    l=$(env $v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "$summ")
    echo "$v -> driver $d | 300 steps $l"
  done
done
