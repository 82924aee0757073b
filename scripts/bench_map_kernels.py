#!/usr/bin/env python3
"""Fixed-M microbench of the map kernels in isolation: M random surfels in front of
the camera (uniform in the frustum, radius from the getRadius rule, PCG-free numpy seed 7), then the
index map, the ACTIVE splat prediction, fuse and clean are timed with HIP events on the launch
stream and priced against the contract's algorithmic bytes (B_idx = 60M + 52N0, B_pred = 60M + 38N0,
B_fuse = 120M + 64N0, B_clean = 120M + 15N0) and the 8 TB/s HBM peak.

    python scripts/bench_map_kernels.py [--surfels 1000000 4000000] [--reps 10]

Prints one JSON line per (M, operator).  Run it under `rocprofv3 --kernel-trace --stats` for the
per-kernel split (profiles/r01_d_map_microbench_*.txt)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def random_map(M, W, H, K, time, rng, in_view=1.0):
    """in_view < 1: the layout of a map grown along a trajectory — the first (1 - in_view) M surfels lie
    outside the frustum (left of it, half of them also older than the 200-frame window), the last
    in_view M are the recent ones, in view and in the scan order of the frames that created them."""
    from densemonoslam_amd import fusion

    fx, fy, cx, cy = K
    s = np.zeros(M, fusion.SURFEL_DTYPE)
    u = rng.uniform(0, W, M).astype(np.float32)
    v = rng.uniform(0, H, M).astype(np.float32)
    z = rng.uniform(0.5, 3.0, M).astype(np.float32)
    n_out = int(round(M * (1.0 - in_view)))
    if n_out:
        u[:n_out] = rng.uniform(-20.0 * W, -0.5 * W, n_out)
        order = np.lexsort((u[n_out:], np.floor(v[n_out:])))  # row-major scan order of the visible part
        u[n_out:], v[n_out:], z[n_out:] = u[n_out:][order], v[n_out:][order], z[n_out:][order]
    s["pos"][:, 0] = (u - cx) * z / fx
    s["pos"][:, 1] = (v - cy) * z / fy
    s["pos"][:, 2] = z
    s["pos"][:, 3] = rng.uniform(0.0, 20.0, M)  # confidence: about half above the threshold 10
    n = rng.normal(0, 0.25, (M, 3)).astype(np.float32)
    n[:, 2] = -1.0  # facing the camera (normals point towards it: -z)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    mean_focal = (fx + fy) / 2.0
    rad = (z / mean_focal) * np.float32(1.41421356237)
    s["nrm"][:, :3] = n
    s["nrm"][:, 3] = np.minimum(2 * rad, rad / np.abs(n[:, 2]))
    col = rng.integers(0, 256, (M, 3))
    s["col"][:, 0] = ((col[:, 0] << 16) + (col[:, 1] << 8) + col[:, 2]).astype(np.float32)
    s["col"][:, 2] = rng.integers(1, time, M)  # init time
    s["col"][:, 3] = time - 1
    s["times"][:] = -3.0
    s["times"][:, 0] = time - 1
    if n_out:
        old = np.arange(n_out) % 2 == 0
        s["times"][:n_out, 0] = np.where(old, time - 400, time - 1)
        s["col"][:n_out, 3] = s["times"][:n_out, 0]
        s["pos"][:n_out, 3] = 15.0  # stable: the clean keeps them
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--surfels", type=int, nargs="+", default=[1_000_000, 4_000_000])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--in-view", type=float, default=1.0,
                    help="fraction of the surfels inside the frustum (1 = the SURVEY 8(d) microbench; 0.05 = a map grown along a trajectory)")
    args = ap.parse_args()
    import torch

    from densemonoslam_amd import fusion, synth

    W, H = args.width, args.height
    K = (528.0, 528.0, 320.0, 240.0)
    N0 = W * H
    time = 1000 if args.in_view < 1.0 else 100
    d, rgb, _ = synth.frame(3, width=W, height=H, K=K, noise=True)
    rgba = synth.rgba(rgb)
    dmf = fusion.depth_metric(fusion.depth_bilateral(d, 3.0), 3.0)
    dm = fusion.depth_metric(d, 3.0)
    pose = fusion.DevicePose(np.eye(4, dtype=np.float32))

    def timed(fn, reps, before=None):
        ms = []
        for _ in range(reps):
            if before:
                before()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        ms.sort()
        return ms[len(ms) // 2]

    for M in args.surfels:
        rng = np.random.default_rng(7)
        surfels = random_map(M, W, H, K, time, rng, args.in_view)
        gm = fusion.GlobalModel(W, H, capacity=M + N0)
        im = fusion.IndexMap(W, H)
        gm.upload(surfels)
        ops = {
            "index_map": (lambda: im.predictIndices(pose, time, 0, gm, K, 25.0, 200), 60.0 * M + 52.0 * N0, None),
            "splat_predict(conf 10)": (lambda: im.combinedPredict(pose, gm, K, 25.0, 10.0, time, 0, time, 200), 60.0 * M + 38.0 * N0, None),
            "splat_predict(conf 0.7)": (lambda: im.combinedPredict(pose, gm, K, 25.0, 0.7, time, 0, time, 200), 60.0 * M + 38.0 * N0, None),
        }
        for name, (fn, nbytes, before) in ops.items():
            ms = timed(fn, args.reps, before)
            print(json.dumps({"M": M, "in_view": args.in_view, "op": name, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                              "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}))
        # fuse and clean change the map: re-upload before every repetition (outside the timed region)
        im.predictIndices(pose, time, 0, gm, K, 25.0, 200)
        ms = timed(lambda: gm.fuse(pose, time, 0, rgba, dm, dmf, im, K, 25.0, 1.0), args.reps, lambda: gm.upload(surfels))
        nbytes = 120.0 * M + 64.0 * N0
        print(json.dumps({"M": M, "in_view": args.in_view, "op": "fuse (associate + in-place update)", "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                          "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                          "note": "the contract prices the reference's whole-map update pass (120 M); this implementation touches <= N0/4 surfels"}))

        def prep_clean():
            gm.upload(surfels)
            im.predictIndices(pose, time, 0, gm, K, 25.0, 200)

        ms = timed(lambda: gm.clean(pose, time, 0, im, K, 10.0, 200, 25.0), args.reps, prep_clean)
        nbytes = 120.0 * M + 15.0 * N0
        print(json.dumps({"M": M, "in_view": args.in_view, "op": "clean (flags + scan + scatter)", "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                          "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                          "surfels_after": gm.lastCount()}))
        gm.close()


if __name__ == "__main__":
    main()
