#!/bin/bash
# On the GPU box: the fixed-M microbench of the map kernels (scripts/bench_map_kernels.py) with, per kernel, the average
# duration (rocprofv3 --kernel-trace) and the bytes it ACTUALLY moved (FETCH_SIZE x 2 + WRITE_SIZE, two PMC passes;
# MI355X_MICROARCH.md HBM section) next to the contract's algorithmic bytes.
# usage: scripts/map_kernel_profile.sh <tag> [bench_map_kernels.py arguments]   -> gpurun_out/map_<tag>/
set -u
tag=$1; shift
out=$PWD/gpurun_out/map_$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
python scripts/bench_map_kernels.py "$@" > $out/microbench.jsonl 2> $out/microbench.err
rocprofv3 --kernel-trace -d $out/kt -o r --output-format csv -- python scripts/bench_map_kernels.py "$@" > /dev/null 2> $out/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o r --output-format csv -- python scripts/bench_map_kernels.py "$@" > /dev/null 2> $out/pmc_$c.err
done
python - "$out" <<'P'
import csv, glob, json, sys, collections
out = sys.argv[1]
def short(n):
    return n.split("(")[0].replace("void dms::", "").replace("dms::", "")
dur = collections.defaultdict(list)
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n.startswith("k_"):
            dur[(n, int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                n = short(r["Kernel_Name"])
                if n.startswith("k_"):
                    acc[(n, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    cnt[c] = acc
rows = []
for key, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 3:
        continue
    us = sorted(v)[len(v) // 2]
    fe = cnt["FETCH_SIZE"].get(key); wr = cnt["WRITE_SIZE"].get(key)
    row = {"kernel": key[0], "grid_threads": key[1], "launches": len(v), "median_us": round(us, 2)}
    if fe and wr:
        kb = 2.0 * sorted(fe)[len(fe) // 2] + sorted(wr)[len(wr) // 2]
        row["actual_MB"] = round(kb / 1024.0, 2)
        row["actual_GBps"] = round(kb * 1024.0 / (us * 1e-6) / 1e9, 1)
    rows.append(row)
with open(out + "/kernels.jsonl", "w") as fh:
    for r in rows:
        fh.write(json.dumps(r) + "\n")
print("\n".join(json.dumps(r) for r in rows[:40]))
P
