#!/bin/bash
# On the GPU box: the map kernels (splat / index project + resolve, clean flags / scatter, associate) on the STREAM the bench
# measures, not on a synthetic map: average duration (rocprofv3 --kernel-trace) and the bytes each launch actually moved
# (2 x FETCH_SIZE + WRITE_SIZE, two PMC passes; MI355X_MICROARCH.md HBM section) next to the contract's bytes for the run's
# surfel count.   usage: scripts/map_kernels_on_stream.sh <tag> [bench.py arguments, e.g. --loop-closure --time-delta 8]
#   -> gpurun_out/stream_<tag>/map_kernels.jsonl
set -u
tag=$1; shift
out=$PWD/gpurun_out/stream_$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
common="--steps 100 --warmup 20 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-pipeline"
rocprofv3 --kernel-trace -d $out/kt -o r --output-format csv -- python bench.py $common "$@" > $out/bench.json 2> $out/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o r --output-format csv -- python bench.py $common "$@" > /dev/null 2> $out/pmc_$c.err
done
python - "$out" <<'P'
import csv, glob, json, sys, collections
out = sys.argv[1]
def short(n):
    return n.split("(")[0].replace("void dms::", "").replace("dms::", "")
dur = collections.defaultdict(list)
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    cnt[c] = acc
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
M, N0 = b["config"]["surfels_per_map"], b["config"]["resolution"][0] * b["config"]["resolution"][1]
contract = {"k_splat_project": 60.0 * M, "k_index_project": 60.0 * M, "k_clean_flags": 60.0 * M + 15.0 * N0, "k_clean_scatter": 60.0 * M,
            "k_index_resolve": 52.0 * N0, "k_splat_resolve": 38.0 * N0, "k_fuse_associate": 64.0 * N0}
rows = []
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    base = k.split("<")[0]
    if base not in contract or len(v) < 20:
        continue
    v = sorted(v[len(v) // 5:])  # the warm-up's launches dropped
    f, w = cnt["FETCH_SIZE"].get(k, []), cnt["WRITE_SIZE"].get(k, [])
    f, w = f[len(f) // 5:], w[len(w) // 5:]
    actual = (2 * sum(f) / max(len(f), 1) + sum(w) / max(len(w), 1)) * 1024.0 if f else None  # FETCH_SIZE counts 32-byte... see guide: KB units x 2 on gfx950
    rows.append({"kernel": k, "launches": len(v), "median_us": round(v[len(v) // 2], 2), "surfels_at_end": M, "contract_MB_at_end": round(contract[base] / 1e6, 2),
                 "actual_MB": None if actual is None else round(actual / 1e6, 2), "actual_over_contract": None if actual is None else round(actual / contract[base], 2)})
with open(out + "/map_kernels.jsonl", "w") as fh:
    fh.write(json.dumps({"stream": sys.argv[1], "frames_per_s_under_rocprof": b["value"], "surfels": M, "note": "single stream (--no-pipeline), 100 steps after 20; actual = 2 x FETCH_SIZE + WRITE_SIZE (KB), averaged over the timed launches; contract bytes use the FINAL surfel count (the map grows during the run: a slight over-estimate of the contract)"}) + "\n")
    for r in rows:
        fh.write(json.dumps(r) + "\n")
        print(json.dumps(r))
P
