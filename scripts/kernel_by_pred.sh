#!/bin/bash
# On the GPU box: average duration of kernels matching $1 in a single-stream bench run, split by the kernel that precedes them
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pipeline > /dev/null 2>&1
python - <<P
import sqlite3,glob,collections
c=sqlite3.connect(glob.glob("/tmp/kt/**/*results.db",recursive=True)[0])
rows=list(c.execute("select name,start,end from kernels order by start"))
d=collections.defaultdict(list)
for i,(n,s,e) in enumerate(rows):
    if "$1" in n:
        prev=rows[i-1][0].split("(")[0][-30:]
        d[(n.split("(")[0][-34:],prev)].append((e-s)/1e3)
for k,v in d.items(): print(k,len(v),round(sum(v)/len(v),2))
P
