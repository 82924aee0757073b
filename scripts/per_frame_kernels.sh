#!/bin/bash
# On the GPU box: single-stream rocprof trace of a short run; kernel time per frame, by kernel, for early / late frames
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps ${1:-20} --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pipeline > /dev/null 2>&1
python - <<P
import sqlite3,glob,collections
c=sqlite3.connect(glob.glob("/tmp/kt/**/*results.db",recursive=True)[0])
rows=list(c.execute("select name,start,end from kernels order by start"))
idx=[i for i,r in enumerate(rows) if "k_so3_level" in r[0]]
def seg(a,b):
    d=collections.defaultdict(float)
    for n,s,e in rows[idx[a]:idx[b]]: d[n.split("(")[0].replace("void ","").replace("dms::","")[:34]]+=(e-s)/1e3/(b-a)
    return d
n=len(idx)
A=seg(1,5); B=seg(n-11,n-1)
print("frames",n,"early sum %.1f late sum %.1f"%(sum(A.values()),sum(B.values())))
for k in sorted(B,key=lambda k:-B[k])[:22]: print("%-36s early %7.2f late %7.2f"%(k,A.get(k,0),B[k]))
P
