#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats of bench.py (pipelined and single-stream) + the two PMC passes of the
# tracker kernels.  usage: scripts/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/{*.txt,*.json}
set -u
tag=$1
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
common="--steps 100 --warmup 10 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg"
rocprofv3 --kernel-trace --stats -d $out/kt -o r -- python bench.py $common > $out/bench_under_rocprof.json 2> $out/kt.err
db=$(find $out/kt -name "*results.db" | head -1)
python scripts/rocprof_summary.py $db 110 > $out/kernel_stats.txt
DMS_NO_PIPELINE=1 rocprofv3 --kernel-trace --stats -d $out/kt1 -o r -- python bench.py $common --no-pipeline > $out/bench_under_rocprof_no_pipeline.json 2> $out/kt1.err
db=$(find $out/kt1 -name "*results.db" | head -1)
python scripts/rocprof_summary.py $db 110 > $out/kernel_stats_no_pipeline.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o r --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg > /dev/null 2> $out/pmc_$c.err
done
python - <<P
import csv, glob, json, collections
out = "$out"
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            name = row["Kernel_Name"].split("(")[0].replace("void dms::", "").replace("dms::", "")
            if "k_gn_level" in name or "k_so3_level" in name:
                acc[name].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[k][c + "_KB_avg"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
json.dump(res, open(out + "/pmc_tracker_kernels.json", "w"), indent=1)
# the record bench.py checks its own nested PMC figure against (copy to profiles/level0_pmc_expected.json with the round's profiles)
l0 = [k for k in res if k.startswith("k_gn_level<true, true,")]
if l0:
    k = max(l0, key=lambda n: int(n.split(",")[2]))
    json.dump({"kernel": k, "hbm_bytes_per_launch": 2 * res[k]["FETCH_SIZE_KB_avg"] * 1024 + res[k]["WRITE_SIZE_KB_avg"] * 1024,
               "source": "profiles/$tag pmc_tracker_kernels.json (2 x FETCH_SIZE + WRITE_SIZE, stand-alone passes of scripts/collect_profiles.sh)"},
              open(out + "/level0_pmc_expected.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
P
head -30 $out/kernel_stats.txt
