#!/usr/bin/env python3
"""Repeated reductions (canonical sums whose first-iteration exponents did not fit) per frame at 1241 x 376 / 40 m, by exp_bias."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densemonoslam_amd import capi, fusion, synth  # noqa: E402

W, H, K = 1241, 376, synth.K_KITTI
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(n)]
for bias in (0, 2, 4, 6, 8):
    ef = fusion.ElasticFusion(W, H, K, model_capacity=3_000_000, depthCut=40.0)
    od = fusion.lib.dms_fusion_odometry(ef.h)
    capi.check(capi.lib.dms_odometry_debug_set(C.c_void_p(od), b"exp_bias", bias))
    t = time.perf_counter()
    total = 0
    tr = capi.TrackResult()
    for d, rgb, _ in frames:
        r = ef.processFrame(rgb, d)
        capi.lib.dms_odometry_fetch_result(C.c_void_p(od), C.byref(tr), None)  # (refreshes the host copy of the tracker's state block)
        one = C.c_int(0)
        capi.check(capi.lib.dms_odometry_canon_retries(C.c_void_p(od), C.byref(one)))
        total += one.value
    el = time.perf_counter() - t
    retries = C.c_int(total)
    print("exp_bias %d: %d repeated reductions in %d frames (%.2f per frame), last pose t = %s, %.1f frames/s (synchronous calls)"
          % (bias, retries.value, n, retries.value / n, np.array(r.pose)[[3, 7, 11]].round(5), n / el))
    ef.close()
