#!/bin/bash
# scripts/ab_args.sh reps "<bench args>" "ENV=a" "ENV=b" ...   (like ab_multi.sh with extra bench.py arguments)
reps=$1; shift; extra=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-kernel-pass --no-full-leg $extra 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(j['value'],1), j['host_blocked_ms_per_step'], j['host_enqueue_ms_per_step'])")
    echo "$v $extra -> $out"
  done
done
