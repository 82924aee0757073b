#!/usr/bin/env python3
"""Frame rate of the real frame step (tracking + fusion, bench.py's stream) when the map is large: the map built
from the first frames of the stream is extended IN FRONT by M_extra stable surfels that lie outside the
frustum (half of them older than the time window) — the situation after a long trajectory, when the camera
looks at a small, recent part of a map of millions of surfels.  Every map pass (index map x2, splat x2,
clean) still streams the whole buffer each frame.

    python scripts/bench_large_map.py [--extra 0 1000000 4000000 16000000] [--steps 200]

One JSON line per size."""
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
    ap.add_argument("--extra", type=int, nargs="+", default=[0, 1_000_000, 4_000_000, 16_000_000])
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    import torch

    from densemonoslam_amd import fusion, synth

    W, H, K = 640, 480, synth.K_640
    dev = torch.device("cuda", 0)
    n_unique = 32
    rgb_t = torch.empty((n_unique, H, W, 3), dtype=torch.uint8, device=dev)
    dep_t = torch.empty((n_unique, H, W), dtype=torch.int16, device=dev)
    for k in range(n_unique):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rgb_t[k] = torch.from_numpy(rgb)
        dep_t[k] = torch.from_numpy(d.view(np.int16))

    def frame_index(i):
        period = 2 * (n_unique - 1)
        j = i % period
        return j if j < n_unique else period - j

    stream = torch.cuda.current_stream().cuda_stream
    for extra in args.extra:
        ef = fusion.ElasticFusion(W, H, K, model_capacity=extra + 4_000_000)
        boot = 20
        for i in range(boot):
            j = frame_index(i)
            ef.processFrameAsync(rgb_t[j].data_ptr(), 3, dep_t[j].data_ptr(), None, 1.0, stream)
        r = ef.fetch(stream)
        if extra:
            gm = ef.globalModel()
            real = gm.downloadMap()
            rng = np.random.default_rng(7)
            far = np.zeros(extra, fusion.SURFEL_DTYPE)
            far["pos"][:, 0] = rng.uniform(-60.0, -8.0, extra)  # metres to the left of the room: never in view
            far["pos"][:, 1] = rng.uniform(-1.0, 1.0, extra)
            far["pos"][:, 2] = rng.uniform(0.5, 3.0, extra)
            far["pos"][:, 3] = 15.0  # stable
            far["nrm"][:, 2] = -1.0
            far["nrm"][:, 3] = 0.005
            far["col"][:, 2] = 1.0
            old = np.arange(extra) % 2 == 0
            far["times"][:] = -3.0
            far["times"][:, 0] = np.where(old, -400.0, 1.0)  # half outside the 200-frame window, half inside
            far["col"][:, 3] = far["times"][:, 0]
            gm.upload(np.concatenate([far, real]))
            del far, real
        for i in range(boot, boot + 10):  # warm-up with the big map
            j = frame_index(i)
            ef.processFrameAsync(rgb_t[j].data_ptr(), 3, dep_t[j].data_ptr(), None, 1.0, stream)
        ef.fetch(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(boot + 10, boot + 10 + args.steps):
            j = frame_index(i)
            ef.processFrameAsync(rgb_t[j].data_ptr(), 3, dep_t[j].data_ptr(), None, 1.0, stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        r = ef.fetch(stream)
        print(json.dumps({"extra_surfels": extra, "map_surfels": int(r.surfels), "frames_per_s": round(args.steps / el, 1),
                          "ms_per_frame": round(1000.0 * el / args.steps, 4), "iterations": list(r.track.iterations_run),
                          "icp_count": float(r.track.lastICPCount)}))
        ef.close()


if __name__ == "__main__":
    main()
