#!/usr/bin/env python3
"""Absolute trajectory error of the frame step against the synthetic stream's ground truth (the per-frame pose error against the
REFERENCE algorithm is zero by construction: the HIP path's poses equal the oracle's bit for bit, tests/test_fusion_gpu.py).
usage: python scripts/ate.py [frames] [width height]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densemonoslam_amd import fusion, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
K = synth.K_640 if (W, H) == (640, 480) else (synth.K_KITTI if (W, H) == (1241, 376) else (0.825 * W, 0.825 * W, W / 2.0, H / 2.0))
ef = fusion.ElasticFusion(W, H, K, model_capacity=8_000_000, depthCut=40.0 if W > 1000 else 3.0)
T0, err_t, err_r, step_t = None, [], [], []
prev_gt = prev_est = None
for k in range(n):
    d, rgb, T = synth.frame(k, width=W, height=H, K=K, noise=True)
    T0 = T if T0 is None else T0
    gt = np.linalg.inv(T0) @ T  # the map's origin is the first camera pose
    r = ef.processFrame(rgb, d)
    est = np.array(r.pose, np.float64).reshape(4, 4)
    err_t.append(np.linalg.norm(est[:3, 3] - gt[:3, 3]))
    Rd = est[:3, :3] @ gt[:3, :3].T
    err_r.append(np.degrees(np.arccos(np.clip((np.trace(Rd) - 1) / 2, -1, 1))))
    if prev_gt is not None:  # relative pose error per step
        dg, de = np.linalg.inv(prev_gt) @ gt, np.linalg.inv(prev_est) @ est
        step_t.append(np.linalg.norm(de[:3, 3] - dg[:3, 3]))
    prev_gt, prev_est = gt, est
err_t, err_r, step_t = np.array(err_t), np.array(err_r), np.array(step_t)
print(json.dumps({"frames": n, "resolution": [W, H], "ate_rmse_mm": round(1000 * float(np.sqrt((err_t ** 2).mean())), 3),
                  "ate_max_mm": round(1000 * float(err_t.max()), 3), "rotation_error_max_deg": round(float(err_r.max()), 4),
                  "relative_translation_error_per_step_rmse_mm": round(1000 * float(np.sqrt((step_t ** 2).mean())), 4),
                  "surfels": int(r.surfels), "what": "frame step on the synthetic box-room stream (depth noise sigma = 1.5 mm z^2, 3 % dropped pixels) against its ground-truth trajectory; "
                  "no alignment (the map's origin is the first pose)"}))
