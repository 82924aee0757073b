"""Golden vectors from the REFERENCE's own pyramid / preparation operators and NID scores (elasticfusion/Core/src/Cuda/cudafuncs.cu).

oracle/ref_build.sh compiles the reference's cudafuncs.cu (renamed CUDA->HIP by the image's hipify-perl at build time; its
texture sampler imageBGRToIntensity removed, nothing else touched) into oracle/_ref/libref_cudafuncs.so in the container that
holds /root/reference; the library travels to the GPU box, where this script runs the reference's operators on an MI355X over
the chain of tests/ref_cases_cf.py and records what they return.

    bash oracle/ref_build.sh                                                  (container with /root/reference)
    gpurun -- 'python tests/golden/make_ref_cudafuncs_golden.py gpurun_out'   (MI355X; writes gpurun_out/ref_cudafuncs.npz)
    cp gpurun_out/ref_cudafuncs.npz tests/golden/ref_cudafuncs.npz            (commit)

tests/test_ref_cf_pin_cpu.py then holds the CPU restatement (oracle/orc_track.c, oracle/orc_nid.py) to these numbers on every
round; tests/test_ref_cf_pin_gpu.py holds the product's operators and its fused pyramid kernels to them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc, ref_cf  # noqa: E402
from tests import ref_cases_cf as cf  # noqa: E402


def main(outdir):
    pair = np.load(os.path.join(ROOT, "tests", "golden", "gputest_pair.npz"))
    z = {}
    for case in ("small", "full"):
        inp = cf.inputs(case, pair, orc)
        out = cf.chain(ref_cf, inp)
        again = cf.chain(ref_cf, inp)  # the reference is deterministic on one device: record that it is
        for k in out:
            assert out[k].tobytes() == again[k].tobytes(), "reference not repeatable: " + k
        for k, v in cf.pack(out).items():
            z["%s_%s" % (case, k)] = v
        for k, v in cf.input_hashes(inp).items():
            z["%s_%s" % (case, k)] = v
        print(case, "nid:", out["nid"])
    z["meta"] = np.array("reference operators: elasticfusion/Core/src/Cuda/cudafuncs.cu via oracle/ref_build.sh (hipify-perl, texture sampler "
                         "lines removed, -ffp-contract=off), gfx950, run on an MI355X")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, "ref_cudafuncs.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
