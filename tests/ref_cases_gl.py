"""Cases that pin the surfel-map half of the hot path (index map, splat prediction, fuse, clean, initialise, depth
pre-filter, fill-in) to the REFERENCE's own GLSL programs (elasticfusion/Core/src/Shaders/*), executed by the image's Mesa llvmpipe
(software OpenGL 4.5) through oracle/ref_gl_harness.c + oracle/ref_gl.py.

One chain of stage calls, three back ends with the same call signatures:
  * GlOps  (oracle.ref_gl: the reference's shaders)  -> tests/golden/make_ref_glsl_golden.py records tests/golden/ref_glsl.npz
                                                       (in the build container: the shader files are read from /root/reference);
  * OrcOps (the restatement, oracle/orc_fusion.c)     -> tests/test_ref_gl_pin_cpu.py, every round on the CPU;
  * HipOps (the product through the C ABI)            -> tests/test_ref_gl_pin_gpu.py.
The state the chain starts from (a surfel map several frames old, the next frame, its pose) comes from a short run of the
restatement's frame loop on the synthetic stream; it is input, hashed into the fixture.  With a `feed` (the fixture) every stage
takes the REFERENCE's recorded output of the previous stage as its input, so each stage is compared in isolation.

What "equal" means here.  GLSL leaves the rounding of a shader's arithmetic to the implementation (no `precise` anywhere in the
reference's shaders): multiply-adds may be fused, exp / pow / inversesqrt / normalize are approximations of implementation-chosen
accuracy, and a point whose centre lies within a sub-pixel step of a pixel boundary may land on either side (rasterisers snap
to 1/256 pixel).  llvmpipe and the NVIDIA driver the reference was developed on differ from each other in exactly these respects,
so the comparison is: every DECISION (which surfels exist, which pixel a surfel lands on, which surfel a pixel associates with,
which surfels are merged / kept / removed, every integer field) identical except where one of the named effects can flip it,
with those cases counted and bounded; every float within a tolerance a few ulp wide (stated per field below).
"""
import hashlib

import numpy as np

W, H = 48, 36
K = (39.6, 39.6, 23.5, 17.5)
MAX_DEPTH = 3.0
N_WARM = 26            # frames of the restatement's frame loop before the snapshot: past tick 20, where the clean's health rule
                       # (copy_unstable.vert:137-150) starts to remove surfels
CONF = 2.5             # confidence threshold used by the stage calls: both sides of it are populated after N_WARM frames
TIME_DELTA = 8         # so that an INACTIVE view exists after N_WARM frames
TEX_DIM = 5700         # GlobalModel::TEXTURE_DIMENSION
STRIDE = 6             # every 6th frame of the synthetic trajectory: the camera leaves parts of the map behind


def configure(**kw):
    """Another image size / warm-up for the live comparison (tests/test_ref_gl_live_cpu.py); returns the previous settings."""
    old = {k: globals()[k] for k in kw}
    globals().update(kw)
    _scale_tolerances()
    return old


def _scale_tolerances():
    """Normals come from a cross product of DIFFERENCES of neighbouring vertices (geometry.glsl:19-39): the differences are one pixel
    footprint (z / fx) long while their rounding errors are ulps of the coordinates (<= 3 m), so the normal's error grows with
    fx: a few ulp(3 m) x fx / z.  2e-5 at the fixture's fx = 39.6 (below), 2.7e-4 at 528, 3.6e-4 at 718.9; radii (depth / |n.z|)
    inherit it."""
    global TOL_NRM, TOL_RAD_REL
    TOL_NRM = 2e-5 * max(1.0, K[0] / 39.6)
    TOL_RAD_REL = TOL_NRM


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1).tobytes()).hexdigest()


# ---- inputs -------------------------------------------------------------------------------------------------------------------
_INPUTS = {}  # one warm-up run per configuration and process: the live test and the pin test of a size share it (26 oracle frames at 1241 x 376 take half a minute)


def inputs(orc, orc_pipeline, synth):
    key = (W, H, tuple(K), N_WARM, STRIDE, MAX_DEPTH, CONF, TIME_DELTA)
    if key not in _INPUTS:
        _INPUTS[key] = _inputs(orc, orc_pipeline, synth)
    return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in _INPUTS[key].items()}


def _inputs(orc, orc_pipeline, synth):
    ef = orc_pipeline.ElasticFusion(W, H, K, timeDelta=TIME_DELTA, confidence=CONF, depthCut=MAX_DEPTH, maxDepthProcessed=MAX_DEPTH)
    ef.depthCut = MAX_DEPTH
    T0 = None
    for k in range(N_WARM):
        d, rgb, T = synth.frame(STRIDE * k, width=W, height=H, K=K, noise=True)
        T0 = T if T0 is None else T0
        ef.processFrame(synth.rgba(rgb), d, inPose=(np.linalg.inv(T0) @ T).astype(np.float32))
    d, rgb, T = synth.frame(STRIDE * N_WARM, width=W, height=H, K=K, noise=True)
    pose = (np.linalg.inv(T0) @ T).astype(np.float32)
    d0, rgb0, _ = synth.frame(0, width=W, height=H, K=K, noise=True)
    # a small deformation graph: nodes sampled from the map (position, identity rotation perturbed, small translation, init time),
    # sorted by time as Deformation::sampleGraphModel leaves them
    rng = np.random.default_rng(5)
    m = ef.model
    sel = m[:: max(len(m) // 24, 1)][:24]
    nodes = np.zeros((len(sel), 16), np.float32)
    nodes[:, 0:3] = sel["pos"][:, :3]
    for i in range(len(sel)):
        a = rng.normal(size=3) * 0.01
        Rm = np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        nodes[i, 3:12] = Rm.T.reshape(9).astype(np.float32)  # column-major 3 x 3
    nodes[:, 12:15] = (rng.normal(size=(len(sel), 3)) * 0.004).astype(np.float32)
    nodes[:, 15] = sel["col"][:, 2]
    nodes = nodes[np.argsort(nodes[:, 15], kind="stable")]
    return dict(depth=d, rgba=synth.rgba(rgb), pose=pose, t_inv=orc.inv4f(pose), model=ef.model.copy(), tick=np.int32(ef.tick),
                depth0=d0, rgba0=synth.rgba(rgb0), nodes=nodes)


def input_hashes(inp):
    return {"in_" + k: np.array(sha(np.asarray(v))) for k, v in inp.items()}


# ---- records <-> arrays -------------------------------------------------------------------------------------------------------
def rec15(model):
    """orc.SURFEL_DTYPE records -> n x 15 floats in the reference's layout (what the fixture stores)."""
    out = np.zeros((len(model), 15), np.float32)
    out[:, 0:4], out[:, 4:8], out[:, 8:11], out[:, 11:15] = model["pos"], model["col"], model["times"][:, :3], model["nrm"]
    return out


def from15(a, dtype):
    a = np.asarray(a, np.float32).reshape(-1, 15)
    out = np.zeros(len(a), dtype)
    out["pos"], out["col"], out["nrm"] = a[:, 0:4], a[:, 4:8], a[:, 11:15]
    out["times"][:, :3] = a[:, 8:11]
    out["times"][:, 3:] = -3.0
    return out


class Rec:
    def __init__(self, feed, dtype):
        self.out, self.feed, self.dtype = {}, feed, dtype

    def __call__(self, name, arr):
        arr = np.ascontiguousarray(arr)
        self.out[name] = arr
        if self.feed is not None and name in self.feed:
            return np.asarray(self.feed[name])
        return arr

    def surfels(self, name, model):
        return from15(self(name, rec15(model)), self.dtype)


T_CONSUME = np.array([[0.9553365, 0.0, 0.29552022, 0.4], [0.0, 1.0, 0.0, -0.1], [-0.29552022, 0.0, 0.9553365, 0.25], [0.0, 0.0, 0.0, 1.0]], np.float32)  # GlobalModel::consume's relativeTransform


# ---- the chain ----------------------------------------------------------------------------------------------------------------
def chain(be, inp, dtype, feed=None, tex_dim=TEX_DIM):
    r = Rec(feed, dtype)
    pose, ti, model, tick = inp["pose"], inp["t_inv"], inp["model"], int(inp["tick"])
    td = TIME_DELTA
    # G1 / G2: ElasticFusion::filterDepth, metriciseDepth (ElasticFusion.cpp:748-768)
    fb = r("bilateral", be.depth_bilateral(inp["depth"], MAX_DEPTH))
    dm = r("metric", be.depth_metric(inp["depth"], MAX_DEPTH))
    dmf = r("metric_f", be.depth_metric(fb, MAX_DEPTH))
    # G3 + G4: first-frame surfels (FeedbackBuffer::compute x 2 + GlobalModel::initialise) from frame 0
    fb0 = be.depth_bilateral(inp["depth0"], MAX_DEPTH) if feed is None else np.asarray(feed["bilateral0"])
    r("bilateral0", fb0)
    boot = r.surfels("boot", be.model_initialise(inp["rgba0"], be.depth_metric(inp["depth0"], MAX_DEPTH), be.depth_metric(fb0, MAX_DEPTH), K, 1, 0,
                                                 float(int(MAX_DEPTH))))
    # GlobalModel::consume (map merge): the frame's map consumes the first frame's surfels moved by a relative transform
    r.surfels("consumed", be.model_consume(model, boot, T_CONSUME))
    # Deformation::sampleGraphModel: every 7th / 64th surfel as {position, init time}, ordered by time
    r("graph_7", be.sample_graph(model, 7))
    r("graph_64", be.sample_graph(model, 64))
    # G5: IndexMap::predictIndices
    im = be.index_map(model, pose, ti, K, H, W, tick, 0, MAX_DEPTH, td)
    im = [r("idx_" + n, a) for n, a in zip(("index", "vertConf", "colorTime", "normRad"), im)]
    # G6: combinedPredict ACTIVE / INACTIVE, synthesizeDepth
    act = be.splat_predict(model, pose, ti, K, H, W, MAX_DEPTH, CONF, tick, 0, tick, td, True)
    act = [r("act_" + n, a) for n, a in zip(("image", "vertex", "normal", "time"), act)]
    ina = be.splat_predict(model, pose, ti, K, H, W, MAX_DEPTH, 0.5, 0, 0, tick - td, td, False)
    [r("ina_" + n, a) for n, a in zip(("image", "vertex", "normal", "time"), ina)]
    lo = be.splat_predict(model, pose, ti, K, H, W, MAX_DEPTH, 0.0, tick, 0, tick, td, True)  # threshold 0: unstable surfels too
    [r("low_" + n, a) for n, a in zip(("image", "vertex", "normal", "time"), lo)]
    dsyn = r("dsyn", be.splat_depth(model, pose, ti, K, H, W, MAX_DEPTH, CONF, tick, 0, tick - td, 65535))
    # G7 + G8: GlobalModel::fuse
    fused, emitted = be.model_fuse(model, pose, tick, 0, inp["rgba"], dm, dmf, im[0], im[1], im[2], im[3], K, MAX_DEPTH, 0.9, tex_dim)
    fused = r.surfels("fused", fused)
    emitted = r.surfels("emitted", emitted)
    # second index map (ElasticFusion.cpp:536) and G9: GlobalModel::clean, without and with a deformation graph
    im2 = be.index_map(fused, pose, ti, K, H, W, tick, 0, MAX_DEPTH, td)
    im2 = [r("idx2_" + n, a) for n, a in zip(("index", "vertConf", "colorTime", "normRad"), im2)]
    r.surfels("cleaned", be.model_clean(fused, emitted, pose, ti, tick, 0, im2[0], im2[1], im2[2], im2[3], K, CONF, td, MAX_DEPTH, None, None, 0))
    r.surfels("cleaned_graph", be.model_clean(fused, emitted, pose, ti, tick, 0, im2[0], im2[1], im2[2], im2[3], K, CONF, td, MAX_DEPTH,
                                              inp["nodes"], dsyn, 0))
    r.surfels("cleaned_fern", be.model_clean(fused, emitted, pose, ti, tick, 0, im2[0], im2[1], im2[2], im2[3], K, CONF, td, MAX_DEPTH,
                                             inp["nodes"], dsyn, 1))
    # G10: FillIn::vertex / normal on the ACTIVE prediction
    r("fill_vertex", be.fill_vertex(act[1], fb, K, False))
    r("fill_normal", be.fill_normal(act[2], fb, K, False))
    r("fill_vertex_pass", be.fill_vertex(act[1], fb, K, True))
    # FillIn::image on the ACTIVE prediction's colours; Resize::image / ::vertex down to W/20 x H/20 (ElasticFusion.cpp:84-97, :443) and to
    # the fern thumbnails' W/8 x H/8 (Ferns.cpp:292-300)
    r("fill_image", be.fill_image(act[0], inp["rgba"], False))
    r("fill_image_pass", be.fill_image(act[0], inp["rgba"], True))
    for nm, (dh, dw) in (("20", (H // 20, W // 20)), ("8", (H // 8, W // 8)), ("odd", (7, 11))):
        r("resize_image_" + nm, be.resize(act[0], dh, dw))
        r("resize_vertex_" + nm, be.resize(act[1], dh, dw))
    return r.out


# ---- back ends ----------------------------------------------------------------------------------------------------------------
class GlOps:
    def __init__(self, ref_gl):
        self.g = ref_gl
        self.depth_bilateral, self.depth_metric, self.model_initialise = ref_gl.depth_bilateral, ref_gl.depth_metric, ref_gl.model_initialise

    def index_map(self, model, pose, ti, K, H, W, time, timeIdx, maxDepth, td):
        return self.g.index_map(model, pose, K, H, W, time, timeIdx, maxDepth, td, t_inv=ti)

    def splat_predict(self, model, pose, ti, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td, active):
        return self.g.splat_predict(model, pose, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td, active, t_inv=ti)

    def splat_depth(self, model, pose, ti, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td):
        return self.g.splat_predict(model, pose, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td, False, depth_only=True, t_inv=ti)

    def model_fuse(self, model, pose, time, timeIdx, rgba, dr, drf, index, vc, ct, nr, K, maxDepth, weighting, tex_dim):
        return self.g.model_fuse(model, pose, time, timeIdx, rgba, dr, drf, index, vc, ct, nr, K, maxDepth, weighting, texDim=tex_dim)

    def model_consume(self, dst, src, T):
        return self.g.model_consume(dst, src, T)

    def sample_graph(self, model, rate):
        return self.g.sample_graph(model, rate)

    def model_clean(self, model, newU, pose, ti, time, timeIdx, index, vc, ct, nr, K, conf, td, maxDepth, nodes, dsyn, isFern):
        return self.g.model_clean(model, newU, pose, time, timeIdx, index, vc, ct, nr, K, conf, td, maxDepth, nodes=nodes, depthSynth=dsyn,
                                  isFern=isFern, t_inv=ti)

    def fill_vertex(self, existing, depth, K, passthrough):
        return self.g.fill(0, existing, depth, K, passthrough)

    def fill_normal(self, existing, depth, K, passthrough):
        return self.g.fill(1, existing, depth, K, passthrough)

    def fill_image(self, existing, rgba, passthrough):
        return self.g.fill_rgb(existing, rgba, passthrough)

    def resize(self, src, drows, dcols):
        return self.g.resize(src, drows, dcols)


class OrcOps:
    def __init__(self, orc):
        self.o = orc
        self.depth_bilateral, self.depth_metric, self.model_initialise = orc.depth_bilateral, orc.depth_metric, orc.model_initialise

    def index_map(self, model, pose, ti, K, H, W, time, timeIdx, maxDepth, td):
        return self.o.index_map(model, pose, K, H, W, time, timeIdx, maxDepth, td)

    def splat_predict(self, model, pose, ti, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td, active):
        return self.o.splat_predict(model, pose, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td, active)

    def splat_depth(self, model, pose, ti, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td):
        return self.o.splat_predict(model, pose, K, H, W, maxDepth, conf, time, timeIdx, maxTime, td, False, depth_only=True)

    def model_fuse(self, model, pose, time, timeIdx, rgba, dr, drf, index, vc, ct, nr, K, maxDepth, weighting, tex_dim):
        m, newU, _ = self.o.model_fuse(model, pose, time, timeIdx, rgba, dr, drf, index, vc, nr, K, maxDepth, weighting)
        return m, newU

    def model_consume(self, dst, src, T):
        return self.o.model_consume(dst, src, T)

    def sample_graph(self, model, rate):
        return self.o.sample_graph(model, rate)

    def model_clean(self, model, newU, pose, ti, time, timeIdx, index, vc, ct, nr, K, conf, td, maxDepth, nodes, dsyn, isFern):
        return self.o.model_clean(model, newU, pose, time, timeIdx, index, vc, ct, K, conf, td, maxDepth, nodes=nodes, depthSynth=dsyn,
                                  isFern=isFern)

    def fill_vertex(self, existing, depth, K, passthrough):
        z = np.zeros(existing.shape[:2] + (4,), np.uint8)
        return self.o.fill_in(existing, existing, z, depth, z, K, passthrough, passthrough)[0]

    def fill_normal(self, existing, depth, K, passthrough):
        z = np.zeros(existing.shape[:2] + (4,), np.uint8)
        return self.o.fill_in(existing, existing, z, depth, z, K, passthrough, passthrough)[1]

    def fill_image(self, existing, rgba, passthrough):
        zf = np.zeros(existing.shape[:2] + (4,), np.float32)
        zd = np.zeros(existing.shape[:2], np.uint16)
        return self.o.fill_in(zf, zf, existing, zd, rgba, K, passthrough, passthrough)[2]

    def resize(self, src, drows, dcols):
        return self.o.resize_nn(src, drows, dcols)


# ---- comparison ---------------------------------------------------------------------------------------------------------------
# Tolerances (absolute unless said otherwise), each a few ulp of the quantity's magnitude in this scene (coordinates <= 3 m):
TOL_POS = 2e-6        # positions, vertex maps: products / sums of 3-4 terms, with or without fused multiply-adds
TOL_NRM = 2e-5        # unit normals: a cross product of differences cancels leading digits before the normalisation
TOL_CONF_REL = 2e-6   # confidences: exp() of the radial weight (an approximation on every GL)
TOL_RAD_REL = 2e-5    # radii: depth / |n.z|, inherits the normal's tolerance
SUBPIXEL = 1.0 / 64   # a point centre this close to a pixel boundary may be rasterised on either side
MAX_AMBIGUOUS = 0.01  # at most this fraction of an index map's pixels may differ through such boundary cases


def _colour_bytes(c):
    c = np.asarray(c, np.float32).astype(np.int64)
    return np.stack([(c >> 16) & 255, (c >> 8) & 255, c & 255], axis=-1)


def compare_surfels(name, got15, want15, exact_values=False, flips_allowed=0):
    """Two n x 15 record arrays of the same stage, record for record."""
    got15, want15 = np.asarray(got15, np.float32).reshape(-1, 15), np.asarray(want15, np.float32).reshape(-1, 15)
    assert got15.shape == want15.shape, "%s: %d records, the reference's shaders produced %d" % (name, len(got15), len(want15))
    if exact_values:
        assert got15.tobytes() == want15.tobytes(), "%s: records differ (expected the same bits)" % name
        return dict(records=len(got15), exact=True)
    g, w = got15, want15
    # integer-valued fields: colour word's other members (0, init time, stamp) and the per-sensor times - exact.  `flips`: the fuse
    # associates a measurement with the NEAREST eligible surfel of its window (data.vert:118-160); where two candidates are within
    # the arithmetic noise of each other the two implementations may pick different ones - two records then differ (one updated
    # here, the other there).  Seen at 1241 x 376 (2 of 655 955 records), never at the smaller sizes; counted, bounded, left out
    # of the value comparison below.
    flip = (g[:, 5:11] != w[:, 5:11]).any(axis=1)
    assert flip.sum() <= flips_allowed, "%s: init time / stamp / sensor times differ in %d records" % (name, int(flip.sum()))
    if flip.any():
        g, w = g[~flip], w[~flip]
    # packed colour: each 8-bit channel within one count (an average that lands on .5 rounds either way)
    dc = np.abs(_colour_bytes(g[:, 4]) - _colour_bytes(w[:, 4]))
    assert dc.max() <= 1, "%s: a colour channel differs by %d" % (name, int(dc.max()))
    assert (dc > 0).any(axis=1).mean() <= 0.01, "%s: %d records with a colour channel off by one" % (name, int((dc > 0).any(axis=1).sum()))
    assert np.abs(g[:, 0:3] - w[:, 0:3]).max() <= TOL_POS, "%s: positions differ by %.3e" % (name, np.abs(g[:, 0:3] - w[:, 0:3]).max())
    assert (np.abs(g[:, 3] - w[:, 3]) <= TOL_CONF_REL * np.maximum(np.abs(w[:, 3]), 1)).all(), "%s: confidences differ by %.3e" % (name, np.abs(g[:, 3] - w[:, 3]).max())
    ok = ~(np.isnan(g[:, 11:14]).any(axis=1) | np.isnan(w[:, 11:14]).any(axis=1))
    assert (np.isnan(g[:, 11:15]) == np.isnan(w[:, 11:15])).all(), "%s: NaN normals in different records" % name
    assert np.abs(g[ok, 11:14] - w[ok, 11:14]).max() <= TOL_NRM, "%s: normals differ by %.3e" % (name, np.abs(g[ok, 11:14] - w[ok, 11:14]).max())
    assert (np.abs(g[ok, 14] - w[ok, 14]) <= TOL_RAD_REL * np.abs(w[ok, 14]) + 1e-9).all(), "%s: radii differ by %.3e" % (name, np.abs(g[ok, 14] - w[ok, 14]).max())
    return dict(records=len(got15), exact=bool(got15.tobytes() == want15.tobytes()), colour_off_by_one=int((dc > 0).any(axis=1).sum()),
                **({"association_flips": int(flip.sum())} if flips_allowed else {}))


def boundary_surfels(model15, t_inv):
    """Surfels whose window coordinates (index_map.vert:49-57, in float64) lie within SUBPIXEL of a pixel boundary: the pixel such a
    point is rasterised on depends on the implementation's sub-pixel snapping."""
    m = np.asarray(model15, np.float64).reshape(-1, 15)
    T = np.asarray(t_inv, np.float64).reshape(4, 4)
    p = m[:, 0:3] @ T[:3, :3].T + T[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        xw = K[0] * p[:, 0] / p[:, 2] + K[2]
        yw = K[1] * p[:, 1] / p[:, 2] + K[3]
    fx, fy = xw - np.floor(xw), yw - np.floor(yw)
    with np.errstate(invalid="ignore"):
        return (fx < SUBPIXEL) | (fx > 1 - SUBPIXEL) | (fy < SUBPIXEL) | (fy > 1 - SUBPIXEL) | ~np.isfinite(xw) | ~np.isfinite(yw)


def compare_index_maps(name, got, want, model15, t_inv):
    """(index, vertConf, colorTime, normRad) x 2.  A pixel may hold different ids only when one of the two surfels is a boundary
    surfel (it was rasterised on this pixel by one implementation and on the neighbouring one by the other), and few pixels may;
    attributes of identical ids to tolerance."""
    gi, wi = np.asarray(got[0]), np.asarray(want[0])
    near = boundary_surfels(model15, t_inv)
    dif = gi != wi
    assert dif.mean() <= MAX_AMBIGUOUS, "%s: ids differ at %.1f %% of the pixels" % (name, 100 * dif.mean())
    unexplained = dif & ~(near[gi] | near[wi])
    assert not unexplained.any(), "%s: ids differ at %d pixels that no boundary case explains" % (name, int(unexplained.sum()))
    same = ~dif
    for k, tol in ((1, TOL_POS), (2, 0.0), (3, TOL_NRM)):
        g, w = np.asarray(got[k])[same], np.asarray(want[k])[same]
        if k == 2:
            assert (g == w).all(), "%s: colorTime differs" % name
        else:  # xyz to tolerance; the fourth member (confidence / radius) is copied
            assert np.abs(g[:, :3] - w[:, :3]).max() <= tol and (g[:, 3] == w[:, 3]).all(), "%s: map %d differs by %.3e" % (name, k, np.abs(g - w).max())
    return dict(pixels_with_a_surfel=int((wi > 0).sum()), ids_differing=int(dif.sum()), boundary_surfels=int(near.sum()))


def compare_splat(name, got, want, coords=None):
    """(image rgba8, vertex, normal, time u16) x 2 of a splat prediction: the same pixels covered, the same surfel winning each
    (colour and time are copied from it), its intersection point / normal to tolerance; pixels where two surfels' depths are
    closer than the arithmetic noise may show the other one."""
    gimg, gv, gn, gt = (np.asarray(a) for a in got)
    wimg, wv, wn, wt = (np.asarray(a) for a in want)
    cov_g, cov_w = gv[..., 2] != 0, wv[..., 2] != 0
    assert (cov_g == cov_w).mean() >= 0.999, "%s: coverage differs at %d pixels" % (name, int((cov_g != cov_w).sum()))
    # the same surfel won: its colour, time stamp and confidence are copies (at full resolution many surfels share a colour and a
    # time, so the confidence - a float accumulated per surfel - is part of the identity)
    same = (gimg == wimg).all(axis=-1) & (gt == wt) & (cov_g == cov_w) & (gv[..., 3] == wv[..., 3])
    assert same.mean() >= 0.995, "%s: another surfel wins at %d pixels" % (name, int((~same).sum()))
    # the fragment's point is the ray / disc intersection k l with k = (p . n) / (l . n) (combo_splat.frag:44-46): its rounding errors
    # are amplified by 1 / |l . n| on discs seen at a grazing angle
    if coords is None:
        hh, ww = gv.shape[:2]
        u, v = np.meshgrid(np.arange(ww, dtype=np.float64) + 0.5, np.arange(hh, dtype=np.float64) + 0.5)
    else:  # sampled form: 1-D lists of pixels, coords = (column, row) of each
        u, v = np.asarray(coords[0], np.float64) + 0.5, np.asarray(coords[1], np.float64) + 0.5
    l = np.stack([(u - K[2]) / K[0], (v - K[3]) / K[1], np.ones_like(u)], axis=-1)
    l /= np.linalg.norm(l, axis=-1, keepdims=True)
    graze = np.maximum(np.abs((l * wn[..., :3].astype(np.float64)).sum(axis=-1)), 0.02)
    dv = np.abs(gv[..., :3].astype(np.float64) - wv[..., :3]).max(axis=-1)
    assert (dv[same] <= TOL_POS / graze[same]).all(), "%s: vertices differ by %.3e (x |l.n| = %.3e)" % (name, dv[same].max(), (dv * graze)[same].max())
    assert np.abs(gn[same] - wn[same]).max() <= TOL_NRM, "%s: normals differ by %.3e" % (name, np.abs(gn[same] - wn[same]).max())
    return dict(covered=int(cov_w.sum()), other_winner=int((~same).sum()), exact_images=bool((gimg == wimg).all() and (gt == wt).all()))


def compare_depth_splat(name, got, want):
    """synthesizeDepth's image (depth_splat.frag): the z of the winning surfel's ray / disc intersection.  Like compare_splat, without
    the attribute maps: a pixel where another surfel wins (two depths closer than the arithmetic noise) shows a different surface
    - rare, counted -, everywhere else the depth agrees to the intersection's tolerance (grazing discs amplify it, see there)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape and ((got != 0) == (want != 0)).mean() >= 0.999, name + ": coverage differs"
    d = np.abs(got - want)
    other = d > 1e-4
    assert other.mean() <= 0.001, "%s: another surfel wins at %d pixels" % (name, int(other.sum()))
    assert d[~other].max() <= 25 * TOL_POS, "%s: differs by %.3e" % (name, d[~other].max())
    return dict(differing=int((d > 0).sum()), above_plain_tolerance=int((d[~other] > TOL_POS).sum()), other_winner=int(other.sum()), worst=float(d[~other].max()))


def compare_image(name, got, want, tol):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, name
    if got.dtype.kind in "iu":
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= tol, "%s: differs by %d" % (name, int(d.max()))
        assert (d > 0).mean() <= 0.002, "%s: %d elements differ" % (name, int((d > 0).sum()))
        return dict(differing=int((d > 0).sum()))
    assert (np.isnan(got) == np.isnan(want)).all(), name + ": NaN masks differ"
    d = np.nan_to_num(np.abs(got.astype(np.float64) - want))
    assert d.max() <= tol, "%s: differs by %.3e (tolerance %.1e)" % (name, d.max(), tol)
    return dict(differing=int((d > 0).sum()), worst=float(d.max()))


def compare_all(out, fx, inp, skip=(), fed=None):
    """Every output of a chain run (`out`) against the fixture / the reference's run (`fx`).  Returns a per-stage summary."""
    rep = {}
    tick, ti = int(inp["tick"]), inp["t_inv"]
    rep["bilateral"] = compare_image("bilateral", out["bilateral"], fx["bilateral"], 1)  # a sum that lands on .5 rounds either way
    rep["bilateral0"] = compare_image("bilateral0", out["bilateral0"], fx["bilateral0"], 1)
    for k in ("metric", "metric_f"):
        assert np.asarray(out[k]).tobytes() == np.asarray(fx[k]).tobytes(), k
    rep["boot"] = compare_surfels("boot", out["boot"], fx["boot"])
    m15 = rec15(inp["model"])
    rep["idx"] = compare_index_maps("idx", [out["idx_" + n] for n in ("index", "vertConf", "colorTime", "normRad")],
                                    [fx["idx_" + n] for n in ("index", "vertConf", "colorTime", "normRad")], m15, ti)
    for pre in ("act", "ina", "low"):
        rep[pre] = compare_splat(pre, [out[pre + "_" + n] for n in ("image", "vertex", "normal", "time")],
                                 [fx[pre + "_" + n] for n in ("image", "vertex", "normal", "time")])
    rep["dsyn"] = compare_depth_splat("dsyn", out["dsyn"], fx["dsyn"])
    n_rec = len(np.asarray(fx["fused"]).reshape(-1, 15))
    flips = 2 * max(1, n_rec // 200000) if n_rec > 100000 else 0  # (one flipped association = two records; none allowed on small maps)
    rep["fused"] = compare_surfels("fused", out["fused"], fx["fused"], flips_allowed=flips)
    # which surfels the frame merged into (their time slot carries the tick): identical sets (up to the flips counted above)
    assert ((np.asarray(out["fused"])[:, 8] == tick) != (np.asarray(fx["fused"])[:, 8] == tick)).sum() <= flips
    rep["fused"]["merged"] = int((np.asarray(fx["fused"])[:, 8] == tick).sum())
    if "emitted" not in skip:
        rep["emitted"] = compare_surfels("emitted", out["emitted"], fx["emitted"])
        rep["emitted"]["new_unstable"] = int((np.asarray(fx["emitted"])[:, 7] == -2).sum())
    rep["idx2"] = compare_index_maps("idx2", [out["idx2_" + n] for n in ("index", "vertConf", "colorTime", "normRad")],
                                     [fx["idx2_" + n] for n in ("index", "vertConf", "colorTime", "normRad")], (fed if fed is not None else fx)["fused"], ti)
    rep["consumed"] = compare_surfels("consumed", out["consumed"], fx["consumed"], exact_values=True)  # one matrix-vector product per record: the same bits
    for k in ("graph_7", "graph_64"):  # copies of map values: the same bytes
        assert np.asarray(out[k], np.float32).tobytes() == np.asarray(fx[k], np.float32).tobytes(), k
        rep[k] = dict(samples=int(len(np.asarray(fx[k]).reshape(-1, 4))), exact=True)
    rep["cleaned"] = compare_surfels("cleaned", out["cleaned"], fx["cleaned"])
    rep["cleaned_graph"] = compare_surfels("cleaned_graph", out["cleaned_graph"], fx["cleaned_graph"])
    rep["cleaned_fern"] = compare_surfels("cleaned_fern", out["cleaned_fern"], fx["cleaned_fern"])
    rep["fill_vertex"] = compare_image("fill_vertex", out["fill_vertex"], fx["fill_vertex"], TOL_POS)
    rep["fill_normal"] = compare_image("fill_normal", out["fill_normal"], fx["fill_normal"], TOL_NRM)
    rep["fill_vertex_pass"] = compare_image("fill_vertex_pass", out["fill_vertex_pass"], fx["fill_vertex_pass"], TOL_POS)
    for k in sorted(fx):  # the copies: FillIn::image (the same bytes), Resize::image / ::vertex (the same texels)
        if k.startswith("fill_image"):
            assert np.asarray(out[k]).tobytes() == np.asarray(fx[k]).tobytes(), k
            rep[k] = dict(exact=True)
        elif k.startswith("resize_"):
            src = (fed if fed is not None else fx)["act_image" if k.startswith("resize_image") else "act_vertex"]  # what both were fed
            rep[k] = compare_resize(k, out[k], fx[k], src)
    return rep


def _resize_texels(n_src, n_dst):
    """Per destination pixel: the source texel floor((j + 0.5) * n_src / n_dst) of a NEAREST fetch at the quad's interpolated
    coordinate, and whether that coordinate lies EXACTLY on a texel boundary (then the rasteriser's rounding of the interpolation
    picks the side: llvmpipe takes the lower row in most of the image and the upper column; the GL specification's exact arithmetic,
    which the restatement follows, the upper texel.  Every row and column is such a tie at the reference's own sizes -
    640 x 480 -> 32 x 24 and 80 x 60)."""
    j = np.arange(n_dst)
    num, den = (2 * j + 1) * n_src, 2 * n_dst
    return num // den, num % den == 0


def compare_resize(name, got, ref, src):
    got, ref, src = np.asarray(got), np.asarray(ref), np.asarray(src)
    assert got.shape == ref.shape and got.dtype == ref.dtype, name
    (tr, tie_r), (tc, tie_c) = _resize_texels(src.shape[0], got.shape[0]), _resize_texels(src.shape[1], got.shape[1])
    as_bytes = lambda a: np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], a.shape[1], -1)
    g, f, s = as_bytes(got), as_bytes(ref), as_bytes(src)
    tie = tie_r[:, None] | tie_c[None, :]
    same = (g == f).all(axis=2)
    assert same[~tie].all(), "%s: differs from the reference's copy away from exact texel boundaries" % name
    for a, who in ((g, "candidate"), (f, "reference")):
        ok = np.zeros(tie.shape, bool)
        for dr in (0, -1):
            for dc in (0, -1):
                allowed = (tie_r[:, None] | (dr == 0)) & (tie_c[None, :] | (dc == 0))
                rr, cc = np.clip(tr + dr, 0, None)[:, None], np.clip(tc + dc, 0, None)[None, :]
                ok |= allowed & (a == s[rr, cc]).all(axis=2)
        assert ok.all(), "%s: the %s's copy holds a texel that is not a neighbour of the sample point" % (name, who)
    return dict(ties=int(tie.sum()), differing_at_ties=int((~same).sum()), pixels=int(tie.size))


# ---- full-size case, recorded as hashes + samples ----------------------------------------------------------------------------
# At 640 x 480 the chain's outputs are 260 MB - not a fixture.  tests/golden/ref_glsl_full.npz therefore holds, per stage,
#   <stage>__orc : SHA-256 of the RESTATEMENT's output in a free run of the chain (no feed) - reproducible anywhere, it is the feed;
#   <stage>__gl  : N_SAMPLES elements (pixels / records, indices from a seeded generator) of what the REFERENCE's SHADERS returned for
#                  that stage when every stage was fed the restatement's outputs - so a candidate fed the same way saw the same inputs.
# compare_sampled applies compare_all's rules to those elements.
N_SAMPLES = 2048
FULL = dict(W=640, H=480, K=(528.0, 528.0, 320.0, 240.0), N_WARM=26, STRIDE=3)
# BASELINE config 4's size with KITTI's intrinsics (KITTI_RGBD_template_params.yaml): 655 955 surfels after 26 frames
KITTI = dict(W=1241, H=376, K=(718.856, 718.856, 607.19, 185.22), N_WARM=26, STRIDE=3)
FULL_CASES = {"640x480": (FULL, "ref_glsl_full.npz"), "1241x376": (KITTI, "ref_glsl_kitti.npz")}
_PIXEL_STAGES = ("bilateral", "metric", "metric_f", "bilateral0", "dsyn", "fill_vertex", "fill_normal", "fill_vertex_pass", "fill_image", "fill_image_pass")
_MAPS = {"idx": ("index", "vertConf", "colorTime", "normRad"), "idx2": ("index", "vertConf", "colorTime", "normRad"),
         "act": ("image", "vertex", "normal", "time"), "ina": ("image", "vertex", "normal", "time"), "low": ("image", "vertex", "normal", "time")}
_RECORD_STAGES = ("boot", "consumed", "fused", "emitted", "cleaned", "cleaned_graph", "cleaned_fern")


def _sample_idx(name, n):
    seed = int(hashlib.sha256(name.encode()).hexdigest()[:8], 16)
    m = min(n, N_SAMPLES)
    return np.sort(np.random.default_rng(seed).choice(n, m, replace=False)) if n else np.zeros(0, np.int64)


def _pix(a, idx):
    a = np.asarray(a)
    return a.reshape(H * W, -1)[idx] if a.ndim == 3 else a.reshape(H * W)[idx]


def sample_outputs(out):
    """The elements of a chain run that the full-size fixture keeps (see above)."""
    z = {}
    pidx = _sample_idx("pixels", H * W)
    for k in _PIXEL_STAGES:
        z[k] = _pix(out[k], pidx)
    for pre, names in _MAPS.items():
        for n in names:
            z[pre + "_" + n] = _pix(out[pre + "_" + n], pidx)
    for k in _RECORD_STAGES:
        a = np.asarray(out[k], np.float32).reshape(-1, 15)
        z[k] = a[_sample_idx(k, len(a))]
        z[k + "__n"] = np.int64(len(a))
    for k in ("graph_7", "graph_64"):
        a = np.asarray(out[k], np.float32).reshape(-1, 4)
        z[k] = a[_sample_idx(k, len(a))]
    return z


def compare_sampled(out, gl, inp, orc_fused15, skip=()):
    """`out`: a candidate's chain run, every stage fed the restatement's outputs; `gl`: sample_outputs() of the shaders' run fed the
    same way (the fixture's <stage>__gl).  compare_all's rules on the sampled elements."""
    rep = {}
    tick, ti = int(inp["tick"]), inp["t_inv"]
    o = sample_outputs(out)
    pidx = _sample_idx("pixels", H * W)
    cols, rows = pidx % W, pidx // W
    rep["bilateral"] = compare_image("bilateral", o["bilateral"], gl["bilateral"], 1)
    rep["bilateral0"] = compare_image("bilateral0", o["bilateral0"], gl["bilateral0"], 1)
    for k in ("metric", "metric_f"):
        assert np.asarray(o[k]).tobytes() == np.asarray(gl[k]).tobytes(), k
    for k in _RECORD_STAGES:
        if k in skip:
            continue
        assert int(o[k + "__n"]) == int(gl[k + "__n"]), "%s: %d records, the reference's shaders produced %d" % (k, int(o[k + "__n"]), int(gl[k + "__n"]))
    rep["boot"] = compare_surfels("boot", o["boot"], gl["boot"])
    m15 = rec15(inp["model"])
    for pre, model15 in (("idx", m15), ("idx2", orc_fused15)):
        rep[pre] = compare_index_maps(pre, [o[pre + "_" + n] for n in _MAPS[pre]], [gl[pre + "_" + n] for n in _MAPS[pre]], model15, ti)
    for pre in ("act", "ina", "low"):
        rep[pre] = compare_splat(pre, [o[pre + "_" + n] for n in _MAPS[pre]], [gl[pre + "_" + n] for n in _MAPS[pre]], coords=(cols, rows))
    rep["dsyn"] = compare_depth_splat("dsyn", o["dsyn"], gl["dsyn"])
    rep["fused"] = compare_surfels("fused", o["fused"], gl["fused"], flips_allowed=2)
    if "emitted" not in skip:
        rep["emitted"] = compare_surfels("emitted", o["emitted"], gl["emitted"])
    rep["consumed"] = compare_surfels("consumed", o["consumed"], gl["consumed"], exact_values=True)
    for k in ("graph_7", "graph_64"):
        assert np.asarray(o[k], np.float32).tobytes() == np.asarray(gl[k], np.float32).tobytes(), k
    for k in ("cleaned", "cleaned_graph", "cleaned_fern"):
        rep[k] = compare_surfels(k, o[k], gl[k])
    rep["fill_vertex"] = compare_image("fill_vertex", o["fill_vertex"], gl["fill_vertex"], TOL_POS)
    rep["fill_normal"] = compare_image("fill_normal", o["fill_normal"], gl["fill_normal"], TOL_NRM)
    rep["fill_vertex_pass"] = compare_image("fill_vertex_pass", o["fill_vertex_pass"], gl["fill_vertex_pass"], TOL_POS)
    for k in ("fill_image", "fill_image_pass"):
        assert np.asarray(o[k]).tobytes() == np.asarray(gl[k]).tobytes(), k
    return rep


def orc_hashes(orc_out):
    return {k + "__orc": np.array(sha(np.asarray(v))) for k, v in orc_out.items()}
