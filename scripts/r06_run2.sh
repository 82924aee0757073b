#!/bin/bash
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_session_gpu.py tests/test_fusion_gpu.py tests/test_ferns_gpu.py tests/test_ref_gl_pin_gpu.py tests/test_collab_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06/pytest_2.txt
cat gpurun_out/r06/pytest_2.txt
quiet="--no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs"
jl() { python -c "
import sys, json
for line in open(sys.argv[1]):
    if line.startswith('{'):
        j = json.loads(line); s = j.get('session') or {}
        print(sys.argv[1].split('/')[-1], round(j['value'], 1), j.get('headline_loop'), 'one_cam', (s.get('one_camera_steady_state') or {}), 'pipelined', {k: v for k, v in (s.get('pipelined') or {}).items() if 'per_s' in k or 'ms_' in k}, 'ag', j.get('allgather_ms_per_frame'), 'fallback', (j.get('fallback_exchange_loop') or {}).get('value'))
" $1; }
for rep in 1 2; do
for fb in 1 0; do
  DMS_SESSION_FUSED_BLOCK=$fb python bench.py --steps 20 --warmup 5 $quiet > gpurun_out/r06/bench_sess_fb${fb}_d.json 2>/dev/null; jl gpurun_out/r06/bench_sess_fb${fb}_d.json
  DMS_SESSION_FUSED_BLOCK=$fb python bench.py --steps 300 --warmup 20 $quiet > gpurun_out/r06/bench_sess_fb${fb}_300.json 2>/dev/null; jl gpurun_out/r06/bench_sess_fb${fb}_300.json
done
done
python bench.py --session-loop --steps 20 --warmup 5 $quiet --no-session-leg > gpurun_out/r06/bench_sl_rccl_d.json 2>/dev/null; jl gpurun_out/r06/bench_sl_rccl_d.json
python bench.py --session-loop --steps 300 --warmup 20 $quiet --no-session-leg > gpurun_out/r06/bench_sl_rccl_300.json 2>/dev/null; jl gpurun_out/r06/bench_sl_rccl_300.json
DMS_SESSION_MAP_STREAMS=0 python bench.py --session-loop --steps 300 --warmup 20 $quiet --no-session-leg > gpurun_out/r06/bench_sl_rccl_300_onestream.json 2>/dev/null; jl gpurun_out/r06/bench_sl_rccl_300_onestream.json
DMS_BENCH_SESSION_TRANSPORT=none python bench.py --session-loop --steps 300 --warmup 20 $quiet --no-session-leg > gpurun_out/r06/bench_sl_local_300.json 2>/dev/null; jl gpurun_out/r06/bench_sl_local_300.json
DMS_BENCH_SESSION_TRANSPORT=none python bench.py --session-loop --steps 20 --warmup 5 $quiet --no-session-leg > gpurun_out/r06/bench_sl_local_d.json 2>/dev/null; jl gpurun_out/r06/bench_sl_local_d.json
