#!/bin/bash
# A/B of library builds on one box: scripts/ab_lib.sh reps libA.so libB.so ...  -> driver form (20 steps after 5) and 300 steps, fps + tracker launch durations
reps=$1; shift
summ='import sys,json; j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); t=j.get("tracker_kernels",{}); print(round(j["value"],1), " ".join("%s %.1f" % (k.replace("gn_level","L").replace("_level",""), v["avg_us"]) for k,v in t.items() if k in ("gn_level0","gn_level1","gn_level2","so3_level")))'
for r in $(seq $reps); do
  for lib in "$@"; do
    d=$(DMS_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "$summ")
    l=$(DMS_LIB_PATH=$PWD/$lib python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-full-leg --no-pmc --no-config-legs --no-session-leg 2>/dev/null | python -c "$summ")
    echo "$lib -> driver $d | 300 steps $l"
  done
done
