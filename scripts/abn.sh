#!/bin/bash
# A/B/C... on one box: scripts/abn.sh reps "<env A>" "<env B>" ...   (bench.py --steps 300, no side passes); prints fps and the level-0 launch duration
reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    out=$(env $v python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); t=j['tracker_kernels']; print(round(j['value'],1), ' '.join('%s %.1f' % (k.replace('gn_level', 'L'), t[k]['avg_us']) for k in ('gn_level0', 'gn_level1', 'gn_level2', 'so3_level', 'track_coarse') if k in t), {l: round(sum(v for k2, v in d.items() if k2 != 'clock_overhead'), 1) for l, d in j.get('gn_level_phase_us_per_frame', {}).items()})")
    echo "$v -> $out"
  done
done
