"""Golden vectors from the REFERENCE's own tracking kernels (elasticfusion/Core/src/Cuda/reduce.cu).

oracle/ref_build.sh compiles the reference's reduce.cu (with its containers/ and headers, renamed CUDA->HIP by the image's
hipify-perl at build time, otherwise untouched) into oracle/_ref/libref_reduce.so in the container that holds
/root/reference; the library travels to the GPU box, where this script runs the reference's icpStep / computeRgbResidual /
rgbStep / so3Step on an MI355X over the cases of tests/ref_cases.py and records what they return.

    bash oracle/ref_build.sh                                               (container with /root/reference)
    gpurun -- 'python tests/golden/make_ref_reduce_golden.py gpurun_out'   (MI355X; writes gpurun_out/ref_reduce.npz)
    cp gpurun_out/ref_reduce.npz tests/golden/ref_reduce.npz              (commit)
    gpurun -- 'DMS_REF_VARIANT=fma python tests/golden/make_ref_reduce_golden.py gpurun_out'   (second fixture, ref_reduce_fma.npz:
                                                                           the reference built with the compiler's default contraction)

tests/test_ref_pin_cpu.py then holds the CPU restatement (oracle/orc_track.c) to these numbers on every round.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc, ref  # noqa: E402
from tests import ref_cases  # noqa: E402


def load_pair():
    z = np.load(os.path.join(ROOT, "tests", "golden", "gputest_pair.npz"))
    return {"rgb1": z["rgb1"], "rgb2": z["rgb2"], "depth1_raw": z["depth1"], "depth2": (z["depth2"] // 5).astype(np.uint16)}


def main_fma(outdir):
    """DMS_REF_VARIANT=fma: the reference built with the compiler's default contraction of device code (libref_reduce_fma.so)
    -> tests/golden/ref_reduce_fma.npz, the comparison point of the restatement's FUSED mode (what the resident tracker runs)."""
    lv = ref_cases.inputs(orc, load_pair())
    base = np.load(os.path.join(ROOT, "tests", "golden", "ref_reduce.npz"))
    out = ref_cases.run(ref, lv, rows_from={k: base[k] for k in ("icp_row_pix", "rgb_row_pix", "rgbres_L2_P1_corres")})
    out = {k: v for k, v in out.items() if not (k.endswith("_valid") or k.endswith("_corres"))}
    out.update(ref_cases.run_trackers(orc, load_pair(), hooks=ref.step_hooks()))
    out.update(ref_cases.input_hashes(lv))
    out["meta"] = np.array("reference kernels: elasticfusion/Core/src/Cuda/reduce.cu via oracle/ref_build.sh, variant built with "
                           "-ffp-contract=fast (the compiler's default for device code), gfx950, run on an MI355X")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, "ref_reduce_fma.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main(outdir):
    if os.environ.get("DMS_REF_VARIANT") == "fma":
        return main_fma(outdir)
    lv = ref_cases.inputs(orc, load_pair())
    out = ref_cases.run(ref, lv)
    again = ref_cases.run(ref, lv, rows_from=out)  # the reference is deterministic on one device: record that it is
    for k in out:
        a, b = out[k], again[k]
        same = (a == b).all() if a.dtype.kind in "iuSU" or a.dtype.names else np.array_equal(a, b, equal_nan=True)
        assert same, "reference not repeatable: " + k
    # whole tracker calls: the restatement's host loop (oracle/orc_odometry.c, plain-order sums) around the reference's kernels
    trk = ref_cases.run_trackers(orc, load_pair(), hooks=ref.step_hooks())
    trk2 = ref_cases.run_trackers(orc, load_pair(), hooks=ref.step_hooks())
    for k in trk:
        assert np.array_equal(trk[k], trk2[k], equal_nan=True), "reference-driven tracker not repeatable: " + k
    out.update(trk)
    out.update(ref_cases.input_hashes(lv))
    out["meta"] = np.array("reference kernels: elasticfusion/Core/src/Cuda/reduce.cu via oracle/ref_build.sh (hipify-perl + hipcc "
                           "-ffp-contract=off, gfx950), run on an MI355X; launch shapes icp/rgb 128x112, residual 256x336, so3 128x64")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, "ref_reduce.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k in sorted(out):
        if k.startswith("icp_L") or k.endswith("_sums") or k.endswith("_iters") or (k.startswith("trk_") and k.endswith("_t")):
            print(k, out[k][-2:])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden"))
