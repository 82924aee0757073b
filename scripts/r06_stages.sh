#!/bin/bash
export PYTHONPATH=$PWD
for v in "$@"; do
  env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        st = j['stage_ms_per_frame']
        print('$v', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 4), 'stages(us):', {k: round(1000 * v, 1) for k, v in st.items()}, 'sum main', round(1000 * sum(v for k, v in st.items() if k not in ('ingest', 'preprocess', 'live_pyramids')), 1))
        sp = j.get('stage_ms_per_frame_pipelined', {})
        print('    pipelined stages(us):', {k: round(1000 * v, 1) for k, v in sp.items()}, 'sum main', round(1000 * sum(v for k, v in sp.items() if k not in ('ingest', 'preprocess', 'live_pyramids')), 1))
        print('    kernels', {k: round(x['avg_us'], 1) for k, x in j['tracker_kernels'].items()}, 'pipelined', {k: round(x['avg_us'], 1) for k, x in j['tracker_kernels_pipelined'].items()})
"
done
