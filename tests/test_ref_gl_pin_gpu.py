"""The HIP path's surfel-map kernels (through the C ABI) against what the REFERENCE's own GLSL programs returned under Mesa's
llvmpipe (tests/golden/ref_glsl.npz, see tests/test_ref_gl_pin_cpu.py): product-vs-reference for SURVEY 8 a8-a14, stage by stage,
each fed the reference's recorded inputs, compared by the same rules as the restatement (tests/ref_cases_gl.compare_all)."""
import numpy as np
import pytest

from tests import ref_cases_gl as cg
from tests.test_ref_gl_pin_cpu import load

pytestmark = pytest.mark.gpu


class HipOps:
    """densemonoslam_amd.fusion behind the chain's call signatures.  The product keeps the fuse's new-unstable buffer inside the map
    object, so model_fuse hands back no `emitted` records (the chain then feeds the fixture's) and model_clean re-runs the fuse on a
    fresh upload before each clean."""

    def __init__(self, fus, dtype, capacity=60000):
        self.f, self.dtype, self.cap = fus, dtype, capacity
        self.W, self.H, self.K = cg.W, cg.H, cg.K  # (the case's size at construction: cg.configure switches it for the full-size case)
        self.gm = fus.GlobalModel(self.W, self.H, capacity=capacity)
        self.gm.setNumSensors(3)  # Shaders/size.glsl: NUM_CAMERAS 3
        self.im = fus.IndexMap(self.W, self.H)
        self.fuse_args = None

    def depth_bilateral(self, d, maxD):
        return self.f.depth_bilateral(d, maxD).download()

    def depth_metric(self, d, maxD):
        return self.f.depth_metric(d, maxD).download()

    def model_initialise(self, rgba, dm, dmf, K_, time, timeIdx, maxDepth):
        gm = self.f.GlobalModel(self.W, self.H, capacity=self.cap)
        gm.initialise(rgba, dm, dmf, K_, time, timeIdx, maxDepth)
        return gm.downloadMap()

    def model_consume(self, dst, src, T):
        a, b = self.f.GlobalModel(self.W, self.H, capacity=2 * self.cap), self.f.GlobalModel(self.W, self.H, capacity=self.cap)
        for m in (a, b):
            m.setNumSensors(3)
        a.upload(dst)
        b.upload(src)
        a.consume(b, T)
        return a.downloadMap()

    def sample_graph(self, model, rate):
        self.gm.upload(model)
        return self.gm.sampleGraph(rate)

    def index_map(self, model, pose, ti, K_, H_, W_, time, timeIdx, maxDepth, td):
        self.gm.upload(model)
        self.im.predictIndices(self.f.DevicePose(pose), time, timeIdx, self.gm, K_, maxDepth, td)
        return self.im.download_index()

    def splat_predict(self, model, pose, ti, K_, H_, W_, maxDepth, conf, time, timeIdx, maxTime, td, active):
        self.gm.upload(model)
        return self.im.combinedPredict(self.f.DevicePose(pose), self.gm, K_, maxDepth, conf, time, timeIdx, maxTime, td, active=active).download()

    def splat_depth(self, model, pose, ti, K_, H_, W_, maxDepth, conf, time, timeIdx, maxTime, td):
        self.gm.upload(model)
        return self.im.synthesizeDepth(self.f.DevicePose(pose), self.gm, K_, maxDepth, conf, time, timeIdx, maxTime, td).download()

    def _maps(self, index, vc, ct, nr):
        self.im.index.upload(np.ascontiguousarray(index, np.uint32))
        self.im.vertConf.upload(np.ascontiguousarray(vc, np.float32))
        self.im.colorTime.upload(np.ascontiguousarray(ct, np.float32))
        self.im.normRad.upload(np.ascontiguousarray(nr, np.float32))

    def _fuse(self):
        model, pose, time, timeIdx, rgba, dr, drf, index, vc, ct, nr, K_, maxDepth, weighting = self.fuse_args
        self.gm.upload(model)
        self._maps(index, vc, ct, nr)
        self.gm.fuse(self.f.DevicePose(pose), time, timeIdx, rgba, dr, drf, self.im, K_, maxDepth, weighting)

    def model_fuse(self, model, pose, time, timeIdx, rgba, dr, drf, index, vc, ct, nr, K_, maxDepth, weighting, tex_dim):
        self.fuse_args = (model, pose, time, timeIdx, rgba, dr, drf, index, vc, ct, nr, K_, maxDepth, weighting)
        self._fuse()
        return self.gm.downloadMap(), np.zeros(0, self.dtype)

    def model_clean(self, model, newU, pose, ti, time, timeIdx, index, vc, ct, nr, K_, conf, td, maxDepth, nodes, dsyn, isFern):
        self._fuse()
        self._maps(index, vc, ct, nr)
        self.gm.clean(self.f.DevicePose(pose), time, timeIdx, self.im, K_, conf, td, maxDepth, graph=nodes, depth_synth=dsyn, isFern=bool(isFern))
        return self.gm.downloadMap()

    def _fill(self, which, existing, depth, K_, passthrough):
        H, W = self.H, self.W
        ex = self.f.PredictionImages(H, W)
        z = np.zeros((H, W, 4), np.float32)
        ex.vertex.upload(np.ascontiguousarray(existing if which == 0 else z, np.float32))
        ex.normal.upload(np.ascontiguousarray(existing if which == 1 else z, np.float32))
        ex.image.upload(np.zeros((H, W, 4), np.uint8))
        out = self.f.fill_in(ex, depth, np.zeros((H, W, 4), np.uint8), K_, passthrough, passthrough)
        return (out.vertex if which == 0 else out.normal).download()

    def fill_vertex(self, existing, depth, K_, passthrough):
        return self._fill(0, existing, depth, K_, passthrough)

    def fill_normal(self, existing, depth, K_, passthrough):
        return self._fill(1, existing, depth, K_, passthrough)

    def fill_image(self, existing, rgba, passthrough):
        H, W = self.H, self.W
        ex = self.f.PredictionImages(H, W)
        z = np.zeros((H, W, 4), np.float32)
        ex.vertex.upload(z)
        ex.normal.upload(z)
        ex.image.upload(np.ascontiguousarray(existing, np.uint8))
        return self.f.fill_in(ex, np.zeros((H, W), np.uint16), rgba, self.K, passthrough, passthrough).image.download()

    def resize(self, src, drows, dcols):
        return self.f.resize_nn(np.ascontiguousarray(src), drows, dcols).download()


def test_product_equals_the_references_shaders(orc):
    from densemonoslam_amd import capi, fusion, synth
    from oracle import orc_pipeline

    assert capi.device_count() >= 1, "no MI355X visible"
    fx = load()
    inp = cg.inputs(orc, orc_pipeline, synth)
    for k, v in cg.input_hashes(inp).items():
        assert str(v) == str(fx[k]), k
    out = cg.chain(HipOps(fusion, orc.SURFEL_DTYPE), inp, orc.SURFEL_DTYPE, feed=fx)
    rep = cg.compare_all(out, fx, inp, skip=("emitted",))
    assert rep["cleaned"]["records"] == len(fx["cleaned"]) and rep["fused"]["merged"] > 100
    print(rep)


@pytest.mark.parametrize("case", ["640x480", "1241x376"])
def test_product_equals_the_references_shaders_at_full_size(orc, case):
    """The same pin at the BASELINE sizes: 640 x 480 (a map of 437 750 surfels) and - round 6 - 1241 x 376 with KITTI's intrinsics (655 955
    surfels); the fuse's update pass wraps TEXTURE_DIMENSION = 5700 many times.  The fixtures keep hashes and samples
    (tests/ref_cases_gl.py "full-size case"): every stage of the product is fed the restatement's outputs - whose hashes the fixture
    pins - exactly as the reference's shaders were when the samples were recorded, and is held (i) to those samples by compare_all's
    rules and (ii) to the restatement's own output of the stage, bit for bit."""
    import os

    from densemonoslam_amd import capi, fusion, synth
    from oracle import orc_pipeline
    from tests.test_ref_gl_pin_cpu import GOLDEN, full_case

    assert capi.device_count() >= 1, "no MI355X visible"
    cfg, name = cg.FULL_CASES[case]
    old = cg.configure(**cfg)
    try:
        fx, inp, orc_out = full_case(orc, orc_pipeline, synth, os.path.join(os.path.dirname(GOLDEN), name))
        gl = {k[:-4]: fx[k] for k in fx.files if k.endswith("__gl")}
        out = cg.chain(HipOps(fusion, orc.SURFEL_DTYPE, capacity=1_400_000), inp, orc.SURFEL_DTYPE, feed=orc_out)
        rep = cg.compare_sampled(out, gl, inp, orc_out["fused"], skip=("emitted",))
        assert rep["fused"]["records"] == cg.N_SAMPLES and int(gl["fused__n"]) > 400_000
        exact = [k for k in out if k != "emitted" and np.asarray(out[k]).tobytes() == np.asarray(orc_out[k]).tobytes()]
        differ = sorted(set(out) - set(exact) - {"emitted"})
        assert not differ, "stages whose product output is not the restatement's, bit for bit: %s" % differ
        print({k: v for k, v in rep.items() if k in ("idx", "act", "fused", "cleaned")})
    finally:
        cg.configure(**old)
