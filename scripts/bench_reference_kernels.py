"""The REFERENCE's own tracking inner loop on this MI355X, beside the product's: oracle/_ref/libref_reduce.so (the reference's
Cuda/reduce.cu built for gfx950, oracle/ref_build.sh) runs computeRgbResidual + icpStep + rgbStep per Gauss-Newton iteration
the way RGBDOdometry.cpp:425-541 does — three kernel pairs, three device synchronisations and three downloads per iteration —
on the GPUTest pair at the three pyramid levels; so3Step likewise.  Host Eigen solve not included (no Eigen in this image).

    gpurun -- 'python scripts/bench_reference_kernels.py > gpurun_out/reference_kernels.json'
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc, ref  # noqa: E402  (test infrastructure: measurement script, not the product)
from tests import ref_cases  # noqa: E402


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "gputest_pair.npz"))
    pair = {"rgb1": z["rgb1"], "rgb2": z["rgb2"], "depth1_raw": z["depth1"], "depth2": (z["depth2"] // 5).astype(np.uint16)}
    lv = ref_cases.inputs(orc, pair)
    lib = ref.lib()
    out = {"what": "reference reduce.cu (hipify-perl + hipcc, gfx950) on this MI355X: microseconds per Gauss-Newton iteration "
                   "(computeRgbResidual + icpStep + rgbStep, each with its own sync + download, RGBDOdometry.cpp:443-541) and per so3Step",
           "levels": {}}
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    R, t = ref_cases.POSES[1]
    total = 0.0
    for lvl, d in enumerate(lv):
        cam = ref_cases._cam(lvl)
        _, kt, krkinv = ref_cases._photo_args(lvl, R, t)
        rows, cols = d["nextImage"].shape
        us_gn, us_so3 = C.c_double(0), C.c_double(0)
        keep = [np.ascontiguousarray(d[k]) for k in ("dIdx", "dIdy", "lastDepth", "nextDepth", "lastImage", "nextImage", "cloud", "vmap_curr",
                                                     "nmap_curr", "vmap_g_prev", "nmap_g_prev")]
        rc = lib.ref_time_iterations(C.c_float(ref_cases.MIN_GRAD[lvl] ** 2 / ref_cases.SOBEL_SCALE ** 2), *[p(a) for a in keep], rows, cols,
                                     p(cam), p(kt), p(krkinv.reshape(9)), 200, 200 if lvl == 2 else 0, C.byref(us_gn), C.byref(us_so3))
        assert rc == 0
        out["levels"]["L%d" % lvl] = {"size": [cols, rows], "us_per_gn_iteration": round(us_gn.value, 2)}
        if lvl == 2:
            out["us_per_so3_iteration"] = round(us_so3.value, 2)
        total += us_gn.value * (10, 5, 4)[lvl]
    total += out["us_per_so3_iteration"] * 10
    out["tracker_call_us_C3"] = round(total, 1)
    out["note"] = ("tracker_call_us_C3 = 10 x L0 + 5 x L1 + 4 x L2 + 10 x so3 iterations (the C3 schedule, all SO3 iterations taken), "
                   "device work and synchronisations only; the product's resident kernels take ~206 us for the same schedule including the "
                   "solves (bench.py `tracker_kernels`)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
