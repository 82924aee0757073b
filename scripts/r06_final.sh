#!/bin/bash
# On the GPU box: the round's closing run - full GPU suite, smoke, the default bench line, the driver's form, kernel stats of the default run.
export PYTHONPATH=$PWD
R=$PWD
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/pytest_final.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r06/pytest_final.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_final.txt 2>&1; tail -2 gpurun_out/r06/smoke_final.txt
python bench.py > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_form_final.json 2> gpurun_out/r06/bench_driver_form_final.err
python - <<'P'
import json
for f in ("bench_final", "bench_driver_form_final"):
    for l in open("gpurun_out/r06/%s.json" % f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f, round(j["value"], 1), j["roofline"]["frac"], j["roofline"]["achieved"], j.get("value_ref_shape"), j.get("value_session_loop"), j["cpu_baseline"]["value"])
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/stats_final -o r --output-format csv -- python $R/bench.py --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg > /dev/null 2> $R/gpurun_out/r06/stats_final.err
ls $R/gpurun_out/r06/stats_final | head
