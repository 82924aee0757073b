#!/bin/bash
# On the GPU box: SQ counters of the kernels whose name contains $1 in any command: scripts/pmc_cmd.sh <name> <command ...>.
# Two passes (the SQ block has 8 counters per pass).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
name=$1; shift
rm -rf /tmp/pm1 /tmp/pm2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d /tmp/pm1 -o r --output-format csv -- "$@" > /dev/null 2> /tmp/pm1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE -d /tmp/pm2 -o r --output-format csv -- "$@" > /dev/null 2> /tmp/pm2.err
python - "$name" <<'P'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for d in ("/tmp/pm1","/tmp/pm2"):
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r["Kernel_Name"]:
                k=r["Kernel_Name"].split("(")[0][-44:]+" grid="+r["Grid_Size"]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d+"/**/*kernel_trace.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0][-44:]+" grid="+r["Grid_Size_X"]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3)
for k,d in acc.items():
    v=dur.get(k,[0]); print(k, "  median us under PMC %.1f (n=%d)"%(sorted(v)[len(v)//2],len(v)))
    for c,vals in d.items(): print("   %-24s avg %.0f  (n=%d)"%(c,sum(vals)/len(vals),len(vals)))
P
tail -2 /tmp/pm1.err /tmp/pm2.err
