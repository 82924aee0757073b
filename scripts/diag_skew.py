"""Arrival skew / completion latency of the two grid-wide reductions of the level-0 resident tracker kernel
(per-block stamps of its middle iteration).  usage: python scripts/diag_skew.py [frames]"""
import ctypes as C
import sys

import numpy as np

from densemonoslam_amd import capi, fusion, synth

W, H, K = 640, 480, (528.0, 528.0, 320.0, 240.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = fusion.ElasticFusion(W, H, K)
od = capi.lib.dms_fusion_odometry(g.h)
for k in range(n):
    d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
    if k == n - 1:
        capi.check(capi.lib.dms_odometry_set_profiling(C.c_void_p(od), 1))
    g.processFrame(rgb, d)
st = np.zeros((256, 8))
ms, cnt = C.c_double(0), C.c_int(0)
for b in range(256):
    for j in range(8):
        capi.check(capi.lib.dms_odometry_get_kernel_time(C.c_void_p(od), ("phase:%d" % (48 + b * 8 + j)).encode(), C.byref(ms), C.byref(cnt)))
        st[b, j] = ms.value * 1e3  # us
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
st[:, :5] -= t0
names = ["pass1 done", "pair complete", "pass2+sum done", "totals complete"]
print("blocks", len(st))
for j in range(4):
    c = st[:, j]
    print("%-16s min %.2f  median %.2f  max %.2f  (spread %.2f us)" % (names[j], c.min(), np.median(c), c.max(), c.max() - c.min()))
print("pair: last arrival -> median completion %.2f us" % (np.median(st[:, 1]) - st[:, 0].max()))
print("sums: last arrival -> median completion %.2f us" % (np.median(st[:, 3]) - st[:, 2].max()))
print("pass 2 + block sum per block: median %.2f us" % np.median(st[:, 2] - st[:, 1]))
by_xcd = [st[i::8, 0].mean() for i in range(8)]
print("pass1-done mean by block%8:", np.round(by_xcd, 2))
print("pair poll start: median %.2f us" % np.median(st[:, 4]))
