#!/usr/bin/env python3
"""Per-step pose difference between the HIP frame step and the CPU oracle over a long teacher-forced run at
the headline resolution (both sides start every step from the GPU's map and pose): the margin under the
north-star bar (1 mm, 0.01 deg per step).  Uses oracle/ as the checker.

    python scripts/parity_margin.py [--frames 100] [--width 640 --height 480]      (round-2 instrument: since round 3 the two sides are bit-identical, tests/test_fusion_gpu.py::test_long_run_every_step_identical_640x480)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    args = ap.parse_args()
    from densemonoslam_amd import fusion, synth
    from oracle import orc, orc_pipeline

    orc.set_threads(min(16, os.cpu_count() or 1))
    W, H = args.width, args.height
    K = synth.K_640 if (W, H) == (640, 480) else (0.825 * W, 0.825 * W, W / 2.0, H / 2.0)
    g = fusion.ElasticFusion(W, H, K, model_capacity=4_000_000)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=4_000_000)
    dts, das, flips = [], [], 0
    for k in range(args.frames):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        pg = np.array(rg.pose, np.float64).reshape(4, 4)
        dt = float(np.linalg.norm(pg[:3, 3] - ro.pose[:3, 3]))
        Rd = pg[:3, :3].T @ ro.pose[:3, :3].astype(np.float64)
        sk = 0.5 * np.array([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]])
        da = float(np.degrees(np.arctan2(np.linalg.norm(sk), (np.trace(Rd) - 1) / 2)))
        if k > 0 and (da > 0.004 or dt > 2e-4):
            print("outlier step %d: dt %.2e m dR %.2e deg | so3 iters %d/%d counts %.0f/%.0f | icp %.0f/%.0f rgb %.0f/%.0f | iters %s/%s" % (
                k, dt, da, rg.track.so3_iterations_run, ro.track.so3_iterations_run, rg.track.lastSO3Count, ro.track.lastSO3Count,
                rg.track.lastICPCount, ro.track.lastICPCount, rg.track.lastRGBCount, ro.track.lastRGBCount,
                list(rg.track.iterations_run), list(ro.track.iterations_run)), file=sys.stderr)
        if k > 0:
            dts.append(dt)
            das.append(da)
            flips += int(bool(rg.fill_in) != ro.fill_in) + int(bool(rg.fused) != ro.fused)
        o.model = g.globalModel().downloadMap()
        o.currPose = np.array(rg.pose, np.float32).reshape(4, 4)
    dts, das = np.array(dts), np.array(das)
    print(json.dumps({"resolution": [W, H], "steps": len(dts), "worst_dt_m": float(dts.max()), "median_dt_m": float(np.median(dts)),
                      "p99_dt_m": float(np.percentile(dts, 99)), "worst_dR_deg": float(das.max()), "median_dR_deg": float(np.median(das)),
                      "p99_dR_deg": float(np.percentile(das, 99)), "decision_flips": flips, "surfels": int(rg.surfels),
                      "bar": {"dt_m": 1e-3, "dR_deg": 0.01},
                      "sums": "canonical"}))


if __name__ == "__main__":
    main()
