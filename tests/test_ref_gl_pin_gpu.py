"""The HIP path's surfel-map kernels (through the C ABI) against what the REFERENCE's own GLSL programs returned under Mesa's
llvmpipe (tests/golden/ref_glsl.npz, see tests/test_ref_gl_pin_cpu.py): product-vs-reference for SURVEY 8 a8-a14, stage by stage,
each fed the reference's recorded inputs, compared by the same rules as the restatement (tests/ref_cases_gl.compare_all)."""
import numpy as np
import pytest

from tests import ref_cases_gl as cg
from tests.test_ref_gl_pin_cpu import load

pytestmark = pytest.mark.gpu
W, H, K = cg.W, cg.H, cg.K


class HipOps:
    """densemonoslam_amd.fusion behind the chain's call signatures.  The product keeps the fuse's new-unstable buffer inside the map
    object, so model_fuse hands back no `emitted` records (the chain then feeds the fixture's) and model_clean re-runs the fuse on a
    fresh upload before each clean."""

    def __init__(self, fus, dtype):
        self.f, self.dtype = fus, dtype
        self.gm = fus.GlobalModel(W, H, capacity=60000)
        self.gm.setNumSensors(3)  # Shaders/size.glsl: NUM_CAMERAS 3
        self.im = fus.IndexMap(W, H)
        self.fuse_args = None

    def depth_bilateral(self, d, maxD):
        return self.f.depth_bilateral(d, maxD).download()

    def depth_metric(self, d, maxD):
        return self.f.depth_metric(d, maxD).download()

    def model_initialise(self, rgba, dm, dmf, K_, time, timeIdx, maxDepth):
        gm = self.f.GlobalModel(W, H, capacity=60000)
        gm.initialise(rgba, dm, dmf, K_, time, timeIdx, maxDepth)
        return gm.downloadMap()

    def model_consume(self, dst, src, T):
        a, b = self.f.GlobalModel(W, H, capacity=60000), self.f.GlobalModel(W, H, capacity=60000)
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
        ex = self.f.PredictionImages(H, W)
        z = np.zeros((H, W, 4), np.float32)
        ex.vertex.upload(z)
        ex.normal.upload(z)
        ex.image.upload(np.ascontiguousarray(existing, np.uint8))
        return self.f.fill_in(ex, np.zeros((H, W), np.uint16), rgba, K, passthrough, passthrough).image.download()

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
