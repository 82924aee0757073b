"""Golden vectors of the CPU oracle on the reference's GPUTest RGB-D pair.

The reference pins no numeric result for this path (its GPUTest harness asserts nothing), so
these vectors pin the ORACLE itself: they were produced by oracle/liborc.so on
tests/golden/gputest_pair.npz with the harness protocol of GPUTest/src/GPUTest.cpp:247-286 and
are checked on every CPU test run (tests/test_oracle_cpu.py) so that an accidental change of the
oracle — the checker of every GPU parity test — cannot go unnoticed.

    python tests/golden/make_oracle_golden.py      (writes tests/golden/oracle_gputest.npz)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from tests import helpers  # noqa: E402

CONFIGS = {
    "C2_icp_fast": dict(rgbOnly=False, icpWeight=100.0, pyramid=False, fastOdom=True, so3=False),
    "C3_full": dict(rgbOnly=False, icpWeight=10.0, pyramid=True, fastOdom=False, so3=True),
    "gputest": dict(rgbOnly=False, icpWeight=10.0, pyramid=False, fastOdom=False, so3=True),
    "rgb_only": dict(rgbOnly=True, icpWeight=10.0, pyramid=True, fastOdom=False, so3=False),
}


def fresh(z):
    K = (528.0, 528.0, 320.0, 240.0)
    verts, norms = helpers.gputest_model_maps(z["depth1"], K)
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    o.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
    o.initRGBModel(helpers.rgba(z["rgb1"]))
    o.initICP((z["depth2"] // 5).astype(np.uint16), 20.0)
    o.initRGB(helpers.rgba(z["rgb2"]))
    o.initFirstRGB(helpers.rgba(z["rgb1"]))
    return o


def compute(z):
    out = {}
    # two forms of the row arithmetic (oracle/orc_track.c): every operation rounded (the product's operator layer; keys
    # without suffix) and multiply-add chains fused in source order (the product's tracker object; keys "_fma")
    for fused, sfx in ((False, ""), (True, "_fma")):
        for name, cfg in CONFIGS.items():
            o = fresh(z)
            o.setFusedRows(fused)
            t, R, res = o.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **cfg)
            out[name + sfx + "_t"] = t
            out[name + sfx + "_R"] = R
            out[name + sfx + "_counts"] = np.array([res.lastICPCount, res.lastRGBCount, res.lastSO3Count, res.so3_iterations_run] + list(res.iterations_run), np.float64)
            out[name + sfx + "_lastA"] = np.array(res.lastA)
    o = fresh(z)
    K = (528.0, 528.0, 320.0, 240.0)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    for lvl in range(3):
        cam = tuple(np.float32(v) / np.float32(1 << lvl) for v in K)
        A, b, r = orc.icpStep(I3, z3, o.buffer(0, lvl), o.buffer(1, lvl), I3, z3, cam, o.buffer(2, lvl), o.buffer(3, lvl), 0.10,
                              float(np.sin(np.radians(20.0))))
        out["icp_A_l%d" % lvl], out["icp_b_l%d" % lvl], out["icp_r_l%d" % lvl] = A, b, r
    d = (z["depth2"] // 5).astype(np.uint16)
    f = orc.depth_bilateral(d, 3.0)
    out["bilateral_checksum"] = np.array([int(f.astype(np.uint64).sum()), int((f > 0).sum()), int(f[240, 320])], np.uint64)
    out["pyr_checksum"] = np.array([int(o.buffer(12, 1).astype(np.uint64).sum()), int(o.buffer(12, 2).astype(np.uint64).sum()),
                                    int(o.buffer(7, 1).astype(np.uint64).sum()), int(o.buffer(7, 2).astype(np.uint64).sum())], np.uint64)
    return out


if __name__ == "__main__":
    z = np.load(os.path.join(ROOT, "tests", "golden", "gputest_pair.npz"))
    out = compute(z)
    path = os.path.join(ROOT, "tests", "golden", "oracle_gputest.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: np.asarray(v).shape for k, v in out.items() if k.endswith("_t")})
