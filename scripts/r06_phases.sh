#!/bin/bash
# phase clocks of the level kernels for a list of environment settings: scripts/r06_phases.sh "ENV=a" "ENV=b" ...
export PYTHONPATH=$PWD
for v in "$@"; do
  env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); t = j['tracker_kernels']
        print('$v', round(j['value'], 1), {k: round(x['avg_us'], 1) for k, x in t.items()})
        for l, d in j.get('gn_level_phase_us_per_frame', {}).items(): print('   ', l, d, round(sum(x for k, x in d.items() if k != 'clock_overhead'), 1))
"
done
