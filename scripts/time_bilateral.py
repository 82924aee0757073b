"""On the GPU box: the depth pre-filter alone (dms_depth_bilateral, whole-chip form), N back-to-back launches on one stream, wall time
per launch; and the same frame filtered twice must give the same bytes.  `python scripts/time_bilateral.py [W H]`"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from densemonoslam_amd import capi, fusion, synth  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
K = (528.0 * W / 640, 528.0 * H / 480, W / 2.0, H / 2.0)
d, _, _ = synth.frame(3, width=W, height=H, K=K, noise=True)
src = capi.DeviceImage.from_array(np.ascontiguousarray(d, np.uint16))
out = fusion.depth_bilateral(src, 3.0)
a = out.download()
lib = capi.lib
import os
MAXD = float(os.environ.get('BIL_MAXD', '3.0'))
for n in (50, 500, 500):
    t0 = time.perf_counter()
    for _ in range(n):
        lib.dms_depth_bilateral(src.ref, out.ref, MAXD, None)
    capi.check(lib.dms_stream_sync(None), "sync")
    dt = (time.perf_counter() - t0) / n
    print("%d x %d: %d launches, %.2f us per launch" % (W, H, n, dt * 1e6))
assert MAXD != 3.0 or 'DMS_BIL_MODE' in os.environ or np.array_equal(a, out.download())
