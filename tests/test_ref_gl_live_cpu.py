"""The restatement's surfel-map half against the REFERENCE's GLSL programs run LIVE by Mesa's llvmpipe (oracle/_ref/libref_gl.so +
the shader files of /root/reference, read at run time), on a larger image and a longer run than the committed fixture's
(tests/test_ref_gl_pin_cpu.py).  Runs on the CPU in the container that holds the reference; skipped anywhere else."""
import numpy as np
import pytest

from tests import ref_cases_gl as cg


@pytest.mark.parametrize("size", [(160, 120, 30, 4), (96, 72, 40, 9), (640, 480, 26, 3), (1241, 376, 26, 3)])
def test_restatement_equals_the_references_shaders_live(orc, size):
    from densemonoslam_amd import synth
    from oracle import orc_pipeline, ref_gl

    if not ref_gl.available():
        pytest.skip("oracle/_ref/libref_gl.so or /root/reference's Shaders/ not here")
    W, H, n_warm, stride = size
    # the two BASELINE sizes with their own intrinsics (GPUTest.cpp:150-152; KITTI_RGBD_template_params.yaml): maps of 437 750 and
    # 655 955 surfels - the fuse's update pass wraps its 5700-wide scratch textures (GlobalModel.cpp:513-694) many times
    Kc = synth.K_640 if (W, H) == (640, 480) else synth.K_KITTI if (W, H) == (1241, 376) else (1.1 * W * 0.75, 1.1 * W * 0.75, W / 2 - 0.5, H / 2 - 0.5)
    old = cg.configure(W=W, H=H, K=Kc, N_WARM=n_warm, STRIDE=stride)
    try:
        inp = cg.inputs(orc, orc_pipeline, synth)
        gl = cg.chain(cg.GlOps(ref_gl), inp, orc.SURFEL_DTYPE)
        out = cg.chain(cg.OrcOps(orc), inp, orc.SURFEL_DTYPE, feed=gl)
        rep = cg.compare_all(out, gl, inp)
        assert rep["cleaned"]["exact"] and rep["fused"]["merged"] > 500 and len(gl["cleaned"]) < len(gl["fused"]) + len(gl["emitted"])
        if W >= 640:
            assert len(gl["fused"]) > 5700 * 50 and rep["fused"].get("association_flips", 0) <= 2
        print(size, {k: v for k, v in rep.items() if k in ("idx", "act", "fused", "cleaned", "cleaned_graph")})
    finally:
        cg.configure(**old)
