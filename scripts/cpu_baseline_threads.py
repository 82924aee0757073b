#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (the oracle/ restatement, bench.py's cpu_baseline leg) on the GPU box's host: frames/s of the
same 640x480 stream at several OpenMP thread counts.  usage: python scripts/cpu_baseline_threads.py [frames] [threads ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from densemonoslam_amd import synth  # noqa: E402
from oracle import orc, orc_pipeline  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
threads = [int(v) for v in sys.argv[2:]] or [1, 8, 16, 32, 64, 128, os.cpu_count() or 1]
W, H, K = 640, 480, synth.K_640
stream = [synth.frame(k, width=W, height=H, K=K, noise=True)[:2] for k in range(32)]


def fi(i):
    j = i % 62
    return j if j < 32 else 62 - j


for t in threads:
    n = orc.set_threads(t)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=8_000_000)
    tot, done = 0.0, 0
    for k in range(frames if t > 1 else min(frames, 12)):
        d, rgb = stream[fi(k)]
        t0 = time.perf_counter()
        o.processFrame(rgb, d)
        dt = time.perf_counter() - t0
        if k >= 2:
            tot += dt
            done += 1
    print(json.dumps({"threads": n, "frames": done, "frames_per_s": round(done / tot, 3), "nproc": os.cpu_count()}), flush=True)
