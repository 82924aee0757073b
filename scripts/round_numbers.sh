#!/bin/bash
# On the GPU box: the bench lines quoted in DESIGN.md §6 for this round -> gpurun_out/r02_final/*.json
out=gpurun_out/r02_final; mkdir -p $out
export PYTHONPATH=$PWD
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --width 1241 --height 376 --steps 200 --no-cpu-baseline --no-kernel-pass > $out/bench_1241x376.json 2>/dev/null
DMS_TRACK_MODE=launches python bench.py --steps 200 --no-cpu-baseline --no-kernel-pass --no-full-leg > $out/bench_launches_mode.json 2>/dev/null
DMS_SHARE_PROJECTION=0 python bench.py --steps 200 --no-cpu-baseline --no-kernel-pass --no-full-leg > $out/bench_no_shared_projection.json 2>/dev/null
python bench.py --loop-closure --time-delta 8 --steps 200 --no-cpu-baseline --no-kernel-pass > $out/bench_full_step_populated.json 2>/dev/null
for f in $out/*.json; do python - <<P
import json
j=json.loads(open("$f").read().strip().split("\n")[-1])
print("$f".split("/")[-1], round(j["value"],1), j.get("roofline",{}).get("frac"), (j.get("full_step") or {}).get("value"))
P
done
