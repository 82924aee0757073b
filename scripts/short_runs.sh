#!/bin/bash
# On the GPU box: the bench line at several --steps (the driver's round-end form is --steps 20 --warmup 5): marginal cost per frame and fixed part
for k in 10 20 40 80; do
  for r in 1 2; do
    python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('steps',j['steps'],'fps',round(j['value'],1),'ms total',round(j['ms_per_step']*j['steps'],3),'blocked',j['host_blocked_ms_per_step'],'M',j['config']['surfels_per_map'])"
  done
done
