#!/usr/bin/env python3
"""Per-frame timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py.

    python scripts/trace_gaps.py gpurun_out/prof6/r6_kernel_trace.csv

Finds the steady-state frames (delimited by k_track_init, launched once per tracked frame), and reports per kernel: launches per
frame, busy time per frame, and the idle gap that precedes it on the device timeline.
"""
import collections
import csv
import sys


def short(n):
    return n.split("(")[0].replace("dms::", "").replace("void ", "")


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    starts = [i for i, e in enumerate(ev) if e[0] in ("k_track_init", "k_so3_level")]  # one per tracked frame
    if len(starts) < 12:
        print("too few frames")
        return
    lo, hi = starts[-11], starts[-1]  # the last 10 complete frames
    frames = 10
    seg = ev[lo:hi]
    wall = seg[-1][2] - seg[0][1]
    busy = collections.defaultdict(float)
    gap = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    prev_end = None
    for n, s, e in seg:
        busy[n] += e - s
        cnt[n] += 1
        if prev_end is not None:
            gap[n] += max(0, s - prev_end)
        prev_end = max(prev_end or e, e)
    tb = sum(busy.values())
    tg = sum(gap.values())
    print("# %d frames: wall %.1f us/frame, kernel busy %.1f us/frame, idle gaps %.1f us/frame, %.1f launches/frame"
          % (frames, wall / frames / 1e3, tb / frames / 1e3, tg / frames / 1e3, len(seg) / frames))
    print("%-40s %8s %10s %10s %10s %10s" % ("kernel", "n/frame", "busy_us/f", "avg_us", "gap_us/f", "avg_gap"))
    for n in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
        print("%-40s %8.1f %10.1f %10.2f %10.1f %10.2f" % (n[:40], cnt[n] / frames, busy[n] / frames / 1e3, busy[n] / cnt[n] / 1e3,
                                                      gap[n] / frames / 1e3, gap[n] / cnt[n] / 1e3))


if __name__ == "__main__":
    main()
