#!/bin/bash
# On the GPU box: kernel trace of the driver's own form of the bench (--steps 20 --warmup 5): per-kernel time over the 20 timed frames
set -u
tag=$1; shift
out=$PWD/gpurun_out/driver_$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
common="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg"
for i in 1 2 3; do python bench.py $common "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('driver form', round(j['value'],1), 'surfels', j['config']['surfels_per_map'])"; done
rocprofv3 --kernel-trace -d $out/kt -o r --output-format csv -- python bench.py $common "$@" > $out/bench.json 2> $out/kt.err
python - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region = the last 20 frames: split on k_live_ingest launches (one per frame)
ing = [i for i, r in enumerate(rows) if "k_live_ingest" in r["Kernel_Name"]]
start = ing[-20] if len(ing) >= 20 else 0
acc = collections.defaultdict(lambda: [0, 0.0])
t0, t1 = int(rows[start]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0].replace("void dms::", "").replace("dms::", "")
    acc[n][0] += 1
    acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
print("# last 20 frames: span %.1f us per frame" % ((t1 - t0) * 1e-3 / 20))
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%-48s %5d launches  %9.1f us per frame  avg %8.2f us" % (n[:48], c, t / 20, t / c))
P
