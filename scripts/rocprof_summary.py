#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) result into a text summary.

    python scripts/rocprof_summary.py gpurun_out/prof1/r1_results.db [frames] > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    frames = float(sys.argv[2]) if len(sys.argv) > 2 else None
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    calls = sum(r[1] for r in rows)
    print("# rocprofv3 --kernel-trace --stats  (durations in microseconds)")
    print("# total kernel time %.1f us over %d launches" % (tot, calls) + ((" = %.1f us and %.1f launches per frame over %g frames" % (tot / frames, calls / frames, frames)) if frames else ""))
    print("%-64s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, n, total, avg, pct in rows:
        short = name.split("(")[0].replace("dms::", "").replace("void ", "")
        print("%-64s %8d %12.1f %10.3f %6.2f%%" % (short[:64], n, total, avg, pct))


if __name__ == "__main__":
    main()
