#!/bin/bash
# round 6, first measurement call: today's driver-form kernel breakdown, the --gpus N headline loop rehearsed on one GPU, the per-level poll-pause sweep
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD
scripts/driver_form_profile.sh r06_base > gpurun_out/r06/driver_form_base.txt 2>&1
quiet="--no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg"
python bench.py --steps 20 --warmup 5 $quiet > gpurun_out/r06/bench_plain_n1.json 2> gpurun_out/r06/bench_plain_n1.err
python bench.py --session-loop --steps 20 --warmup 5 $quiet > gpurun_out/r06/bench_session_loop_n1.json 2> gpurun_out/r06/bench_session_loop_n1.err
python bench.py --session-loop --steps 300 --warmup 20 $quiet > gpurun_out/r06/bench_session_loop_n1_300.json 2>> gpurun_out/r06/bench_session_loop_n1.err
DMS_BENCH_SESSION_TRANSPORT=none python bench.py --session-loop --steps 300 --warmup 20 $quiet > gpurun_out/r06/bench_session_loop_n1_300_local.json 2>> gpurun_out/r06/bench_session_loop_n1.err
python bench.py --steps 300 --warmup 20 $quiet > gpurun_out/r06/bench_plain_n1_300.json 2>> gpurun_out/r06/bench_plain_n1.err
DMS_BENCH_SESSION=0 DMS_BENCH_SHARE_GPU=1 DMS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 $quiet > gpurun_out/r06/bench_two_gloo_ranks.json 2> gpurun_out/r06/bench_two_gloo_ranks.err
for f in plain_n1 session_loop_n1 session_loop_n1_300 session_loop_n1_300_local plain_n1_300 two_gloo_ranks; do
  python - $f <<'P'
import json, sys
f = sys.argv[1]
try:
    j = json.loads(open("gpurun_out/r06/bench_%s.json" % f).read().strip().split("\n")[-1])
    print(f, round(j["value"], 1), j.get("headline_loop"), "allgather_ms", j.get("allgather_ms_per_frame"), "rccl_ranks", j.get("rccl_ranks"), j.get("rccl_library"),
          "fallback", (j.get("fallback_exchange_loop") or {}).get("value"), "err", (j.get("session_loop") or {}).get("error"))
except Exception as e:
    print(f, "FAILED", e)
P
done
scripts/abn.sh 2 "X=0" "DMS_AR_FIRST_DELAY_BY_LEVEL=-1,-1,4,4" "DMS_AR_FIRST_DELAY_BY_LEVEL=-1,-1,12,12" "DMS_AR_FIRST_DELAY_BY_LEVEL=-1,-1,16,16" "DMS_AR_FIRST_DELAY_BY_LEVEL=-1,16,-1,-1" "DMS_AR_FIRST_DELAY_BY_LEVEL=-1,32,-1,-1" "DMS_AR_FIRST_DELAY_BY_LEVEL=16,-1,-1,-1" "DMS_AR_FIRST_DELAY_BY_LEVEL=32,-1,-1,-1" > gpurun_out/r06/first_delay_sweep.txt 2>&1
cat gpurun_out/r06/first_delay_sweep.txt
tail -40 gpurun_out/r06/driver_form_base.txt
