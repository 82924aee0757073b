#!/bin/bash
# On the GPU box: single-stream rocprof kernel summaries of bench.py under two environments.  usage: scripts/kt_compare.sh "<env A>" "<env B>"
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/ktc; env $v rocprofv3 --kernel-trace --stats -d /tmp/ktc -o r -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-pipeline > /dev/null 2>&1
  echo "== $v"; python scripts/rocprof_summary.py $(find /tmp/ktc -name "*results.db" | head -1) 110 | head -22
done
