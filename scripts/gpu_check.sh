#!/bin/bash
# usage (on the GPU box): scripts/gpu_check.sh <tag> [pytest-args...]   -> gpurun_out/r02/{pytest,bench}_<tag>.*
tag=$1; shift
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -8 > gpurun_out/r02/pytest_$tag.txt
cat gpurun_out/r02/pytest_$tag.txt
python bench.py --steps 100 --warmup 10 > gpurun_out/r02/bench_$tag.json 2> gpurun_out/r02/bench_$tag.err
python - <<P
import json
j=json.loads(open("gpurun_out/r02/bench_$tag.json").read().strip().split("\n")[-1])
print("fps", round(j["value"],1), "enq", j["host_enqueue_ms_per_step"], "blocked", j["host_blocked_ms_per_step"], "full", round(j["full_step"]["value"],1), "roofline", round(j["roofline"]["frac"],4))
print({k:v["avg_us"] for k,v in j["tracker_kernels"].items()})
for l,d in j["gn_level_phase_us_per_frame"].items(): print(l, d)
print(j["stage_ms_per_frame"])
P
