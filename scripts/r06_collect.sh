#!/bin/bash
# round 6: the records that go to profiles/ (run on the GPU box through gpurun)
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD TMPDIR=/tmp
scripts/collect_profiles.sh r06 > gpurun_out/r06/collect.log 2>&1
scripts/driver_form_profile.sh r06 > gpurun_out/r06/driver_form_kernels.txt 2>&1
python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_form.json 2> gpurun_out/r06/bench_driver_form.err
python bench.py --width 1241 --height 376 --no-cpu-baseline --no-pmc > gpurun_out/r06/bench_1241x376.json 2>/dev/null
scripts/driver_form_repeat.sh 5 > gpurun_out/r06/driver_form_x5.json 2>/dev/null
quiet="--no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg"
python bench.py --session-loop --steps 20 --warmup 5 $quiet > gpurun_out/r06/bench_session_loop_one_rank_rccl.json 2>/dev/null
python bench.py --session-loop --steps 300 --warmup 20 $quiet > gpurun_out/r06/bench_session_loop_one_rank_rccl_300.json 2>/dev/null
DMS_BENCH_SESSION=1 DMS_BENCH_SHARE_GPU=1 DMS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 $quiet > gpurun_out/r06/bench_two_gloo_ranks_one_gpu.json 2> gpurun_out/r06/bench_two_gloo_ranks_one_gpu.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r06/kt_sess -o r -- python bench.py --session-loop --steps 100 --warmup 10 $quiet > /dev/null 2> gpurun_out/r06/kt_sess.err
db=$(find gpurun_out/r06/kt_sess -name "*results.db" | head -1)
python scripts/rocprof_summary.py $db 110 > gpurun_out/r06/session_loop_kernel_stats.txt
rm -rf gpurun_out/r06/kt_sess
tail -3 gpurun_out/r06/collect.log; head -30 gpurun_out/r06/driver_form_kernels.txt; cat gpurun_out/r06/driver_form_x5.json; head -40 gpurun_out/r06/session_loop_kernel_stats.txt
