#!/bin/bash
# On the GPU box: SQ counters of one kernel (substring $1) in a single-stream 20-step bench run
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d /tmp/pm -o r --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pipeline > /dev/null 2> /tmp/pm.err
python - <<P
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "$1" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in d.items(): print("   %-24s avg %.0f  (n=%d)"%(c,sum(v)/len(v),len(v)))
P
tail -3 /tmp/pm.err
