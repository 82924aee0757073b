#!/usr/bin/env python3
"""Main-stream timeline of the frame step from a rocprofv3 --kernel-trace CSV of bench.py (pipelined run): for the last
frames, every main-stream kernel's average duration and the average idle gap in front of it on ITS stream (the prep stream's
kernels — ingest, depth filter, live pyramids — are listed apart).  usage: python scripts/main_stream_gaps.py <kernel_trace.csv> [frames]"""
import collections
import csv
import sys

PREP = ("k_live_ingest", "k_depth_bilateral", "k_live_levels")


def short(n):
    return n.split("(")[0].replace("dms::", "").replace("void ", "")


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
    main_ev = [e for e in ev if not e[2].startswith(PREP) and e[2].startswith("k_")]
    starts = [i for i, e in enumerate(main_ev) if e[2] == "k_so3_level"]
    lo, hi = starts[-nf - 1], starts[-1]
    seg = main_ev[lo:hi]
    wall = (seg[-1][1] - seg[0][0]) / nf / 1e3
    dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    order = []
    prev_end = None
    for s, e, n in seg:
        if n not in dur:
            order.append(n)
        dur[n] += (e - s) / 1e3
        cnt[n] += 1
        if prev_end is not None:
            gap[n] += (s - prev_end) / 1e3
        prev_end = e
    print("# %d frames: %.1f us per frame on the main stream = %.1f busy + %.1f idle" % (nf, wall, sum(dur.values()) / nf, sum(gap.values()) / nf))
    print("%-44s %8s %10s %12s" % ("kernel (in launch order)", "n/frame", "avg_us", "gap_before_us"))
    for n in order:
        print("%-44s %8.2f %10.2f %12.2f" % (n[:44], cnt[n] / nf, dur[n] / cnt[n], gap[n] / cnt[n]))
    prep = [e for e in ev if e[2].startswith(PREP)]
    d = collections.defaultdict(list)
    for s, e, n in prep[-3 * nf:]:
        d[n].append((e - s) / 1e3)
    for n, v in d.items():
        print("prep  %-38s %8.2f %10.2f" % (n[:38], len(v) / nf, sum(v) / len(v)))


if __name__ == "__main__":
    main()
