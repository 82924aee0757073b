"""Golden vectors from the REFERENCE's own GLSL programs (elasticfusion/Core/src/Shaders/*.vert, *.geom, *.frag).

oracle/ref_build.sh compiles oracle/ref_gl_harness.c into oracle/_ref/libref_gl.so: an OpenGL host (context on the image's Mesa
llvmpipe, textures, buffers, uniforms, draw calls restating IndexMap.cpp / GlobalModel.cpp / ComputePack.cpp / FeedbackBuffer.cpp /
FillIn.cpp) that loads the reference's shader files from /root/reference at RUN time and lets Mesa's GLSL compiler build them.
This script drives the chain of tests/ref_cases_gl.py through it and records what the reference's shaders returned.  It runs on
the CPU, in the container that holds /root/reference (no GPU involved):

    bash oracle/ref_build.sh
    python tests/golden/make_ref_glsl_golden.py            (writes tests/golden/ref_glsl.npz; commit it)
    python tests/golden/make_ref_glsl_golden.py --full        (640 x 480: tests/golden/ref_glsl_full.npz, hashes + samples)
    python tests/golden/make_ref_glsl_golden.py --full-kitti  (1241 x 376, KITTI intrinsics: tests/golden/ref_glsl_kitti.npz)

tests/test_ref_gl_pin_cpu.py then holds the CPU restatement (oracle/orc_fusion.c) to these numbers on every round;
tests/test_ref_gl_pin_gpu.py holds the product's kernels to them on the MI355X.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from densemonoslam_amd import synth  # noqa: E402  (the synthetic stream: host-side numpy only)
from oracle import orc, orc_pipeline, ref_gl  # noqa: E402
from tests import ref_cases_gl as cg  # noqa: E402


def main(path):
    inp = cg.inputs(orc, orc_pipeline, synth)
    out = cg.chain(cg.GlOps(ref_gl), inp, orc.SURFEL_DTYPE)
    again = cg.chain(cg.GlOps(ref_gl), inp, orc.SURFEL_DTYPE)  # llvmpipe rasterises in parallel tiles: record that the results do not depend on it
    for k in out:
        assert out[k].tobytes() == again[k].tobytes(), "not repeatable: " + k
    # the literal texture filter of the reference's host code (LINEAR on the RGB / raw metric depth textures, see ref_gl_harness.c):
    # recorded as a fact about llvmpipe, not used by any comparison
    ref_gl.lib().rgl_set_linear(1)
    dm = ref_gl.depth_metric(inp["depth0"], cg.MAX_DEPTH)
    n_lin = len(ref_gl.vertex_feedback(inp["rgba0"], dm, cg.K, 1, 0, cg.MAX_DEPTH, True))
    ref_gl.lib().rgl_set_linear(0)
    n_near = len(ref_gl.vertex_feedback(inp["rgba0"], dm, cg.K, 1, 0, cg.MAX_DEPTH, True))
    z = dict(out)
    z.update(cg.input_hashes(inp))
    z["feedback_count_linear_vs_nearest"] = np.array([n_lin, n_near], np.int64)
    progs = [("empty.vert", "quad.geom", f) for f in ("fill_rgb.frag", "resize.frag", "depth_bilateral.frag", "depth_metric.frag", "fill_vertex.frag",
                                                      "fill_normal.frag")]
    progs += [("index_map.vert", "", "index_map.frag"), ("splat.vert", "", "combo_splat.frag"), ("splat.vert", "", "depth_splat.frag"),
              ("data.vert", "data.geom", "data.frag"), ("update.vert", "", ""), ("copy_unstable.vert", "copy_unstable.geom", ""),
              ("init_unstable.vert", "", ""), ("vertex_feedback.vert", "vertex_feedback.geom", "")]
    ok = [ref_gl.try_program(*p)[0] for p in progs]
    assert all(ok), [p for p, o in zip(progs, ok) if not o]
    z["meta"] = np.array("reference GLSL programs (elasticfusion/Core/src/Shaders) compiled and run by %s through oracle/ref_gl_harness.c; programs built: %s"
                         % (ref_gl.renderer(), ", ".join("+".join(x for x in p if x) for p in progs)))
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path), "bytes;", str(z["meta"])[:120])
    print("surfels emitted by vertex_feedback with LINEAR / NEAREST filtering of the raw depth:", n_lin, n_near)


def main_full(path, case="640x480"):
    """640 x 480 / 1241 x 376 (tests/ref_cases_gl.py "full-size case"): hashes of the restatement's free run (the feed) + samples of what
    the reference's shaders return stage by stage on that feed + the whole-array comparison report of this recording run."""
    cg.configure(**cg.FULL_CASES[case][0])
    inp = cg.inputs(orc, orc_pipeline, synth)
    orc_out = cg.chain(cg.OrcOps(orc), inp, orc.SURFEL_DTYPE)
    gl_out = cg.chain(cg.GlOps(ref_gl), inp, orc.SURFEL_DTYPE, feed=orc_out)
    again = cg.chain(cg.GlOps(ref_gl), inp, orc.SURFEL_DTYPE, feed=orc_out)
    for k in gl_out:
        assert gl_out[k].tobytes() == again[k].tobytes(), "not repeatable: " + k
    rep = cg.compare_all(orc_out, gl_out, inp, fed=orc_out)  # (a free run of the restatement IS the restatement fed its own outputs)
    z = {k + "__gl": v for k, v in cg.sample_outputs(gl_out).items()}
    z.update(cg.input_hashes(inp))
    z.update(cg.orc_hashes(orc_out))
    z["report"] = np.array(str(rep))
    z["meta"] = np.array("reference GLSL programs (elasticfusion/Core/src/Shaders) compiled and run by %s through oracle/ref_gl_harness.c at %d x %d; %d samples "
                         "per stage; every stage fed the restatement's outputs (hashes: <stage>__orc)" % (ref_gl.renderer(), cg.W, cg.H, cg.N_SAMPLES))
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k, v in rep.items():
        print(" ", k, v)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("--full", "--full-kitti"):
        case = "640x480" if sys.argv[1] == "--full" else "1241x376"
        main_full(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", cg.FULL_CASES[case][1]), case)
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref_glsl.npz"))
