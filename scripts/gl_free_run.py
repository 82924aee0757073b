#!/usr/bin/env python3
"""Free-running fusion-only comparison: the REFERENCE's GLSL programs (oracle/ref_gl.py: Mesa llvmpipe runs the shader files of
/root/reference) and the restatement (oracle/orc_fusion.c) each build their OWN surfel map over the same frames at the same given
poses - no stage is fed the other side's output - and the script reports, per frame, how far the two maps have drifted apart:
surfel counts, surfels merged / appended / removed by the frame on either side, and the two maps' ACTIVE model views compared
pixel by pixel.  A decision that differs (a boundary pixel associating with another surfel, a record kept on one side and removed on
the other) shifts every later surfel id, so ids are not compared; what the tracker and the next fuse consume - the views - is.

Runs on the CPU in the container that holds /root/reference.   python scripts/gl_free_run.py [W H frames stride] > profiles/..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from densemonoslam_amd import synth  # noqa: E402
from oracle import orc, orc_pipeline, ref_gl  # noqa: E402
from tests import ref_cases_gl as cg  # noqa: E402

W, H, FRAMES, STRIDE = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (320, 240, 30, 3)))
K = synth.K_640 if (W, H) == (640, 480) else (0.825 * W, 0.825 * W, W / 2.0, H / 2.0)
MAXD, CONF, TD = 3.0, 10.0, 200
cg.configure(W=W, H=H, K=K)
orc.set_threads(min(16, os.cpu_count() or 1))
sides = {"shaders": cg.GlOps(ref_gl), "restatement": cg.OrcOps(orc)}
maps = {}
print("# %d x %d, %d frames (every %d. of the synthetic trajectory), given poses, confidence %.0f, time window %d" % (W, H, FRAMES, STRIDE, CONF, TD))
print("# frame | surfels shaders / restatement | merged | new points flagged unstable | not kept by the clean (map + emitted records) | ACTIVE view (conf 0.7): covered shaders / restatement, "
      "coverage differs, |dz| > 1 mm | stable view (conf 10): covered, coverage differs, |dz| > 1 mm")
T0 = None
for k in range(FRAMES):
    d, rgb, T = synth.frame(STRIDE * k, width=W, height=H, K=K, noise=True)
    T0 = T if T0 is None else T0
    pose = (np.linalg.inv(T0) @ T).astype(np.float32)
    ti = orc.inv4f(pose)
    rgba = synth.rgba(rgb)
    tick = k + 1
    stats, views = {}, {}
    for name, be in sides.items():
        fb = be.depth_bilateral(d, MAXD)
        dm, dmf = be.depth_metric(d, MAXD), be.depth_metric(fb, MAXD)
        if k == 0:
            maps[name] = be.model_initialise(rgba, dm, dmf, K, tick, 0, float(int(MAXD)))
            stats[name] = (len(maps[name]), 0, 0, 0)
        else:
            m = maps[name]
            im = be.index_map(m, pose, ti, K, H, W, tick, 0, MAXD, TD)
            fused, emitted = be.model_fuse(m, pose, tick, 0, rgba, dm, dmf, im[0], im[1], im[2], im[3], K, MAXD, 1.0, cg.TEX_DIM)
            merged = int((fused["times"][:, 0] == tick).sum())
            im2 = be.index_map(fused, pose, ti, K, H, W, tick, 0, MAXD, TD)
            cleaned = be.model_clean(fused, emitted, pose, ti, tick, 0, im2[0], im2[1], im2[2], im2[3], K, CONF, TD, MAXD, None, None, 0)
            n_new = int((emitted["col"][:, 3] == -2).sum()) if len(emitted) else 0
            stats[name] = (len(cleaned), merged, n_new, len(fused) + len(emitted) - len(cleaned))
            maps[name] = cleaned
        views[name] = [be.splat_predict(maps[name], pose, ti, K, H, W, MAXD, c, tick, 0, tick, TD, True)[1] for c in (0.7, CONF)]
    a, b = stats["shaders"], stats["restatement"]
    row = "%3d | %7d / %7d | %6d / %6d | %5d / %5d | %5d / %5d |" % (k, a[0], b[0], a[1], b[1], a[2], b[2], a[3], b[3])
    for i in range(2):
        va, vb = views["shaders"][i][..., 2], views["restatement"][i][..., 2]
        ca, cb = va != 0, vb != 0
        both = ca & cb
        row += " %6d / %6d, %4d, %4d |" % (int(ca.sum()), int(cb.sum()), int((ca != cb).sum()), int((np.abs(va - vb)[both] > 1e-3).sum()))
    print(row, flush=True)
