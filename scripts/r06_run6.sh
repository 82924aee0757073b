#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r06
timeout 1700 python -m pytest tests/test_refframe_gpu.py tests/test_ferns_gpu.py tests/test_session_gpu.py tests/test_tracking_gpu.py tests/test_cpp_mirror.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06/pytest_6.txt
cat gpurun_out/r06/pytest_6.txt
for v in "DMS_TRACK_LONG_RESIDENT=0" "DMS_TRACK_LONG_RESIDENT=1"; do
  env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); s = j['session']
        print('$v', round(j['value'], 1), 'sync: before', s['frames_per_s_before_merge'], 'ms/tick', s['ms_per_tick_before_merge'], 'merge tick ms', s['ms_merge_tick'], 'after', s['frames_per_s_after_merge'], '| pipelined before', s['pipelined']['frames_per_s_before_merge'], 'merge tick', s['pipelined']['ms_merge_tick'], 'after', s['pipelined']['frames_per_s_after_merge'], 'merges', s['merges'], s['pipelined']['merges'])
"
done
