#!/usr/bin/env python3
"""Aggregate throughput of several camera contexts on ONE GPU (collaborative sessions with more
cameras than GPUs): C cameras, each with its own map, tracker and stream, frames enqueued round
robin without host synchronisation.  Not the headline metric (bench.py, one camera per GPU) — it
shows how much of the chip a single latency-bound camera leaves idle.

    python scripts/bench_multicam.py [--cameras 1 2 4] [--steps 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cameras", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--threads", action="store_true", help="one host thread per camera (each blocks on its own run-ahead bound)")
    args = ap.parse_args()
    from densemonoslam_amd import capi, fusion, synth

    W, H, K = 640, 480, (528.0, 528.0, 320.0, 240.0)
    n_unique = 16
    for C_ in args.cameras:
        cams, streams, bufs = [], [], []
        for c in range(C_):
            frames = [synth.frame(k, cam_id=c, width=W, height=H, K=K, noise=True) for k in range(n_unique)]
            rb = [capi.DeviceBuffer(W * H * 3).upload(np.ascontiguousarray(f[1], np.uint8)) for f in frames]
            db = [capi.DeviceBuffer(W * H * 2).upload(np.ascontiguousarray(f[0], np.uint16)) for f in frames]
            bufs.append((rb, db))
            cams.append(fusion.ElasticFusion(W, H, K, model_capacity=4_000_000, timeIdx=c, num_sensors=max(3, c + 1)))
            streams.append(capi.create_stream())

        def idx(i):
            period = 2 * (n_unique - 1)
            j = i % period
            return j if j < n_unique else period - j

        def run_cam(c, lo, hi):
            for i in range(lo, hi):
                cams[c].processFrameAsync(bufs[c][0][idx(i)].ptr, 3, bufs[c][1][idx(i)].ptr, None, 1.0, streams[c])
            capi.check(capi.lib.dms_stream_sync(streams[c]))

        def run(lo, hi):
            if args.threads and C_ > 1:
                import threading

                th = [threading.Thread(target=run_cam, args=(c, lo, hi)) for c in range(C_)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                return
            for i in range(lo, hi):
                for c in range(C_):
                    cams[c].processFrameAsync(bufs[c][0][idx(i)].ptr, 3, bufs[c][1][idx(i)].ptr, None, 1.0, streams[c])
            for c in range(C_):
                capi.check(capi.lib.dms_stream_sync(streams[c]))

        run(0, args.warmup)
        t0 = time.perf_counter()
        run(args.warmup, args.warmup + args.steps)
        dt = time.perf_counter() - t0
        surf = [int(cams[c].fetch(streams[c]).surfels) for c in range(C_)]
        print(json.dumps({"cameras_on_one_gpu": C_, "frames_per_s_aggregate": round(C_ * args.steps / dt, 1),
                          "frames_per_s_per_camera": round(args.steps / dt, 1), "ms_per_round": round(1000 * dt / args.steps, 4), "surfels": surf,
                          "host_threads": C_ if args.threads else 1, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                          "DMS_PERSIST_MAX_BLOCKS": os.environ.get("DMS_PERSIST_MAX_BLOCKS"), "DMS_PERSIST_UNCHAINED": os.environ.get("DMS_PERSIST_UNCHAINED")}))
        for c in cams:
            c.close()
        for s in streams:
            capi.destroy_stream(s)


if __name__ == "__main__":
    main()
