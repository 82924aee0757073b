#!/bin/bash
# The driver's form of the bench (--steps 20 --warmup 5: 631 088 surfels, most of them younger than the clean's 20-frame rule) N times over,
# fresh process each: N x 20 timed steps at the SAME surfel count - the long-run figure to put beside the 300-step leg, whose map has
# shrunk to its stable 182 k surfels by then.  usage: scripts/driver_form_repeat.sh N [env...]
n=$1; shift
export PYTHONPATH=$PWD
for i in $(seq $n); do
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null
done | python -c "
import sys, json
v, m = [], set()
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); v.append(j['value']); m.add(j['config']['surfels_per_map'])
print(json.dumps({'form': 'python bench.py --steps 20 --warmup 5, %d fresh processes' % len(v), 'timed_steps_total': 20 * len(v), 'surfels_per_map': sorted(m),
                  'frames_per_s_mean': round(sum(v) / len(v), 1), 'min': round(min(v), 1), 'max': round(max(v), 1), 'ms_per_step_mean': round(sum(1000.0 / x for x in v) / len(v), 4)}))
"
