#!/bin/bash
# scripts/stage_scan.sh VAR v1 v2 ... : bench with VAR=v, print fps + preprocess stage + level-0 kernel time
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-leg 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('$var=$v', round(j['value'],1), 'preprocess', j['stage_ms_per_frame']['preprocess'], 'L0', round(j['tracker_kernels']['gn_level0']['avg_us'],1))"
done
