#!/bin/bash
# A/B/C... on one box: scripts/abn.sh reps "<env A>" "<env B>" ...   (bench.py --steps 300, no side passes); prints fps and the level-0 launch duration
reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); t=j['tracker_kernels']; print(round(j['value'],1), 'L0', round(t['gn_level0']['avg_us'],1), 'L1', round(t['gn_level1']['avg_us'],1), 'L2', round(t['gn_level2']['avg_us'],1), 'so3', round(t['so3_level']['avg_us'],1))")
    echo "$v -> $out"
  done
done
