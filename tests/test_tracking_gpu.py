"""GPU parity tests of the tracking half: HIP (through the C ABI) vs the CPU oracle.

Bars (BASELINE.json north_star):
  * integer / byte / index outputs and per-pixel fp32 outputs: bit-exact (both sides evaluate
    the same fp32 expression order with contraction off and IEEE div/sqrt);
  * reduced sums (A, b, residual): fp32 tree sum on the GPU vs fp64 sum in the oracle —
    relative tolerance 2e-4 of the matrix scale, written at each assert;
  * recovered pose: <= 1 mm and <= 0.01 degree per tracking call.
"""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dms():
    from densemonoslam_amd import capi, odometry

    assert capi.device_count() >= 1, "no MI355X visible"
    return odometry


def _sum_close(a, b, rtol=2e-4, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "%s: max |Δ| / max|ref| = %.3e > %.1e" % (what, err, rtol)


# ------------------------------------------------------------------------------------------
# pyramid / preparation kernels: bit-exact
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def prep_inputs(gputest_pair):
    d = gputest_pair["depth2"]
    rgb = helpers.rgba(gputest_pair["rgb2"])
    return d, rgb, gputest_pair["K"]


def test_pyrDown_exact(dms, orc, prep_inputs):
    d, _, _ = prep_inputs
    g = dms.ops.pyrDown(d)
    assert (g.download() == orc.pyrDown(d)).all()
    g2 = dms.ops.pyrDown(g)
    assert (g2.download() == orc.pyrDown(orc.pyrDown(d))).all()


def test_vmap_nmap_exact(dms, orc, prep_inputs):
    d, _, K = prep_inputs
    for cutoff in (20.0, 1.2):
        vg = dms.ops.createVMap(K, d, cutoff)
        vo = orc.createVMap(K, d, cutoff)
        assert helpers.planes_equal_where_valid(vg.download(), vo)
        ng = dms.ops.createNMap(vg)
        no = orc.createNMap(vo)
        assert helpers.planes_equal_where_valid(ng.download(), no)


def test_copy_resize_transform_exact(dms, orc, gputest_pair):
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], gputest_pair["K"])
    vg, ng = dms.ops.copyMaps(verts, norms)
    vo, no = orc.copyMaps(verts, norms)
    assert helpers.nan_equal(vg.download(), vo) and helpers.nan_equal(ng.download(), no)
    v1g, n1g = dms.ops.resizeVMap(vg), dms.ops.resizeNMap(ng)
    v1o, n1o = orc.resizeMap(vo, False), orc.resizeMap(no, True)
    assert helpers.planes_equal_where_valid(v1g.download(), v1o)
    assert helpers.planes_equal_where_valid(n1g.download(), n1o)
    R = np.array([[0.9998, -0.0175, 0.01], [0.0174, 0.9998, 0.012], [-0.0102, -0.0118, 0.99988]], np.float32)
    t = np.array([0.01, -0.02, 0.03], np.float32)
    tvg, tng = dms.ops.tranformMaps(v1g, n1g, R, t)
    tvo, tno = orc.tranformMaps(v1o, n1o, R, t)
    assert helpers.planes_equal_where_valid(tvg.download(), tvo)
    assert helpers.planes_equal_where_valid(tng.download(), tno)
    only_v = dms.ops.tranformMaps(v1g, None, R, t)
    assert helpers.planes_equal_where_valid(only_v.download(), orc.tranformMaps(v1o, None, R, t))
    # copyMaps single-map overload
    assert helpers.nan_equal(dms.ops.copyMaps(verts, None).download(), orc.copyMaps(verts, None))


def test_intensity_pyramids_sobel_exact(dms, orc, prep_inputs):
    _, rgba, _ = prep_inputs
    ig = dms.ops.imageBGRToIntensity(rgba)
    io = orc.imageBGRToIntensity(rgba)
    assert (ig.download() == io).all()
    i1g = dms.ops.pyrDownUcharGauss(ig)
    i1o = orc.pyrDownUcharGauss(io)
    assert (i1g.download() == i1o).all()
    i2g = dms.ops.pyrDownUcharGauss(i1g)
    assert (i2g.download() == orc.pyrDownUcharGauss(i1o)).all()
    # an image with holes (zeros are skipped; an all-zero window gives 0)
    holes = io.copy()
    holes[100:140, 200:260] = 0
    assert (dms.ops.pyrDownUcharGauss(holes).download() == orc.pyrDownUcharGauss(holes)).all()
    for img in (io, i1o):
        dxg, dyg = dms.ops.computeDerivativeImages(img)
        dxo, dyo = orc.computeDerivativeImages(img)
        assert (dxg.download() == dxo).all() and (dyg.download() == dyo).all()


def test_float_depth_paths_exact(dms, orc, gputest_pair):
    verts, _ = helpers.gputest_model_maps(gputest_pair["depth1_raw"], gputest_pair["K"])
    K = gputest_pair["K"]
    dg = dms.ops.verticesToDepth(verts, 1.5)
    do = orc.verticesToDepth(verts, 1.5)
    assert helpers.nan_equal(dg.download(), do)
    d1g = dms.ops.pyrDownGaussF(dg)
    d1o = orc.pyrDownGaussF(do)
    assert helpers.nan_equal(d1g.download(), d1o)
    assert helpers.nan_equal(dms.ops.pyrDownGaussF(d1g).download(), orc.pyrDownGaussF(d1o))
    for lvl, (g, o) in enumerate(((dg, do), (d1g, d1o))):
        assert helpers.nan_equal(dms.ops.projectToPointCloud(g, K, lvl).download(), orc.projectToPointCloud(o, K, lvl))
    vo, _ = orc.copyMaps(verts, None), None
    assert helpers.nan_equal(dms.ops.verticesToDepth2D(vo, 1.5).download(), do)


def test_ragged_sizes_exact(dms, orc):
    """Odd sizes (1241×376-style halves) and a tiny image exercise every border branch."""
    rng = np.random.default_rng(5)
    for (h, w) in ((94, 155), (47, 77), (16, 18)):
        d = rng.integers(0, 4000, (h, w)).astype(np.uint16)
        d[rng.random((h, w)) < 0.2] = 0
        assert (dms.ops.pyrDown(d).download() == orc.pyrDown(d)).all()
        K = (200.0, 210.0, w / 2.0, h / 2.0)
        vg, vo = dms.ops.createVMap(K, d, 3.0), orc.createVMap(K, d, 3.0)
        assert helpers.planes_equal_where_valid(vg.download(), vo)
        assert helpers.planes_equal_where_valid(dms.ops.createNMap(vg).download(), orc.createNMap(vo))
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        assert (dms.ops.pyrDownUcharGauss(img).download() == orc.pyrDownUcharGauss(img)).all()
        dx, dy = dms.ops.computeDerivativeImages(img)
        ox, oy = orc.computeDerivativeImages(img)
        assert (dx.download() == ox).all() and (dy.download() == oy).all()
        f = rng.random((h, w)).astype(np.float32) * 5
        f[rng.random((h, w)) < 0.3] = np.nan
        assert helpers.nan_equal(dms.ops.pyrDownGaussF(f).download(), orc.pyrDownGaussF(f))


# ------------------------------------------------------------------------------------------
# reduction operators
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tracker_pair(dms, orc, gputest_pair):
    """Both trackers initialised with the GPUTest protocol (GPUTest.cpp:247-286) plus the RGB
    initialisation the harness omits (so the photometric terms are defined)."""
    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    rgba1, rgba2 = helpers.rgba(gputest_pair["rgb1"]), helpers.rgba(gputest_pair["rgb2"])
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    pose = np.eye(4, dtype=np.float32)
    for trk in (g, o):
        trk.initICPModel(verts, norms, 20.0, pose)
        trk.initRGBModel(rgba1)
        trk.initICP(gputest_pair["depth2"], 20.0)
        trk.initRGB(rgba2)
        trk.initFirstRGB(rgba1)
    return g, o


def test_odometry_pyramids_exact(tracker_pair):
    g, o = tracker_pair
    for lvl in range(3):
        for which in (0, 1, 2, 3):
            assert helpers.planes_equal_where_valid(g.buffer(which, lvl), o.buffer(which, lvl)), (which, lvl)
        for which in (4, 5):
            assert helpers.nan_equal(g.buffer(which, lvl), o.buffer(which, lvl)), (which, lvl)
        for which in (6, 7, 8, 12):
            assert (g.buffer(which, lvl) == o.buffer(which, lvl)).all(), (which, lvl)


@pytest.mark.parametrize("use_b", [False, True])
def test_fused_model_pyramid_equals_operator_chain(dms, orc, gputest_pair, use_b):
    """dms_odometry_initModelFused (the frame step's 4-launch model pyramid: device-side source
    selection, pose from HBM) against the oracle's initICPModel + initRGBModel on the selected
    source, bit for bit (planes compared where the x plane is not NaN, as everywhere else)."""
    K = gputest_pair["K"]
    vA, nA = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    rng = np.random.default_rng(5)
    vB = vA.copy()
    vB[..., :3] += rng.normal(0, 0.01, vB[..., :3].shape).astype(np.float32)
    vB[rng.random(vB.shape[:2]) < 0.05] = 0.0  # holes: z == 0 => NaN vertex and normal
    nB = np.roll(nA, 3, axis=1).copy()
    iA, iB = helpers.rgba(gputest_pair["rgb1"]), helpers.rgba(gputest_pair["rgb2"])
    a = 0.05
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    pose[:3, 3] = (0.1, -0.2, 0.05)
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    g.initModelFused(vA, nA, iA, vB, nB, iB, use_b, False, pose)
    o.initICPModel(vB if use_b else vA, nB if use_b else nA, 20.0, pose)
    o.initRGBModel(iB if use_b else iA)
    for lvl in range(3):
        for which in (2, 3):
            assert helpers.planes_equal_where_valid(g.buffer(which, lvl), o.buffer(which, lvl)), (which, lvl)
        assert helpers.nan_equal(g.buffer(4, lvl), o.buffer(4, lvl)), ("lastDepth", lvl)
        assert (g.buffer(6, lvl) == o.buffer(6, lvl)).all(), ("lastImage", lvl)
    # frameToFrameRGB: the image alone comes from B
    g.initModelFused(vA, nA, iA, vB, nB, iB, False, True, pose)
    o.initRGBModel(iB)
    for lvl in range(3):
        assert (g.buffer(6, lvl) == o.buffer(6, lvl)).all(), ("lastImage forced", lvl)


def test_nid_scores_parity(dms, gputest_pair):
    """NID key-framing operators: joint histograms integer-exact against the numpy
    restatement of the reference's kernels, scores within 1e-5 (parallel fp64 sums on the GPU, the
    reference's scalar float loops in the oracle)."""
    from oracle import orc_nid

    rng = np.random.default_rng(11)
    g1 = helpers.rgba(gputest_pair["rgb1"])[..., :3].astype(np.float32) @ np.array([0.114, 0.299, 0.587], np.float32)
    g2 = helpers.rgba(gputest_pair["rgb2"])[..., :3].astype(np.float32) @ np.array([0.114, 0.299, 0.587], np.float32)
    img_kf, img_curr = g1.astype(np.uint8), g2.astype(np.uint8)
    img_old = np.roll(img_kf, 5, axis=1)
    d_kf = gputest_pair["depth1_raw"].astype(np.float32) / 5000.0
    d_kf[d_kf == 0] = np.nan
    d_old = d_kf + rng.normal(0, 0.05, d_kf.shape).astype(np.float32)
    d_old[rng.random(d_kf.shape) < 0.3] = np.nan
    d_cur = gputest_pair["depth2"].astype(np.float32) / 1000.0
    d_cur[d_cur == 0] = np.nan
    d_cur[::7, ::5] = 30.0  # beyond max_depth: clamped into the last bin
    for bins in (64, 16):
        s_g, h_g = dms.ops.computeNIDImg(img_kf, img_old, d_kf, d_old, img_curr, bins)
        s_o, h_o = orc_nid.nid_img(img_kf, img_old, d_kf, d_old, img_curr, bins)
        assert (h_g == h_o).all() and int(h_g.sum()) == img_kf.size
        assert abs(s_g - float(s_o)) < 1e-5, (s_g, s_o)
        assert 0.0 < s_g <= 1.0
    for bins, maxd in ((500, 25000.0), (100, 3000.0)):
        s_g, h_g = dms.ops.computeNIDDepth(d_kf, d_old, d_cur, bins, maxd)
        s_o, h_o = orc_nid.nid_depth(d_kf, d_old, d_cur, bins, maxd)
        assert (h_g == h_o).all() and int(h_g.sum()) == d_kf.size
        assert abs(s_g - float(s_o)) < 1e-5, (s_g, s_o)
    # identical frames: the joint histogram is diagonal, MI = H, nid = 0
    ones = np.ones_like(d_kf)  # a prediction everywhere (a pixel without one contributes intensity 0)
    s_same, _ = dms.ops.computeNIDImg(img_kf, img_kf, ones, ones, img_kf, 64)
    assert abs(s_same) < 1e-6


def test_icpStep_parity(dms, orc, tracker_pair):
    g, o = tracker_pair
    K = (528.0, 528.0, 320.0, 240.0)
    R = np.eye(3, dtype=np.float32)
    t = np.zeros(3, np.float32)
    for lvl in range(3):
        cam = tuple(np.float32(v) / np.float32(1 << lvl) for v in K)
        bufs = [o.buffer(w, lvl) for w in (0, 1, 2, 3)]
        Ao, bo, ro = orc.icpStep(R, t, bufs[0], bufs[1], R, t, cam, bufs[2], bufs[3], 0.10, float(np.sin(np.radians(20.0))))
        Ag, bg, rg = dms.ops.icpStep(R, t, bufs[0], bufs[1], R, t, cam, bufs[2], bufs[3], 0.10, float(np.sin(np.radians(20.0))))
        assert rg[1] == ro[1], "inlier count must be exact (level %d): %r vs %r" % (lvl, rg[1], ro[1])
        _sum_close(Ag, Ao, what="A level %d" % lvl)
        _sum_close(bg, bo, rtol=2e-3, what="b level %d" % lvl)  # signed terms cancel: looser than A
        _sum_close(rg[0], ro[0], what="residual level %d" % lvl)
        assert np.allclose(Ag, Ag.T)
    # explicit launch shape (reference-style threads/blocks arguments) gives the same sums
    bufs = [o.buffer(w, 0) for w in (0, 1, 2, 3)]
    A1, b1, r1 = dms.ops.icpStep(R, t, bufs[0], bufs[1], R, t, K, bufs[2], bufs[3], 0.10, 0.342, threads=256, blocks=112)
    A2, b2, r2 = dms.ops.icpStep(R, t, bufs[0], bufs[1], R, t, K, bufs[2], bufs[3], 0.10, 0.342)
    assert r1[1] == r2[1]
    _sum_close(A1, A2)


def test_rgb_residual_and_step_parity(dms, orc, tracker_pair):
    g, o = tracker_pair
    K = (528.0, 528.0, 320.0, 240.0)
    sobelScale = 1.0 / 8.0
    for lvl, minGrad in ((0, 5.0), (1, 3.0), (2, 1.0)):
        cam = [np.float32(v) / np.float32(1 << lvl) for v in K]
        nextImage, lastImage = o.buffer(7, lvl), o.buffer(6, lvl)
        nextDepth, lastDepth = o.buffer(5, lvl), o.buffer(4, lvl)
        dx, dy = orc.computeDerivativeImages(nextImage)
        Km = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float64)
        ang = 0.01
        Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        krkinv = (Km @ Rm @ np.linalg.inv(Km)).astype(np.float32)
        kt = (Km @ np.array([0.004, -0.002, 0.003])).astype(np.float32)
        minScale = float(minGrad ** 2 / sobelScale ** 2)
        co, so, no = orc.computeRgbResidual(minScale, dx, dy, lastDepth, nextDepth, lastImage, nextImage, 0.07, kt, krkinv)
        cg, sg, ng = dms.ops.computeRgbResidual(minScale, dx, dy, lastDepth, nextDepth, lastImage, nextImage, 0.07, kt, krkinv)
        assert (sg, ng) == (so, no), "count / Σdiff² are integers: exact (level %d)" % lvl
        cgh = cg.download()
        for f in ("zero_x", "zero_y", "one_x", "one_y", "diff", "valid"):
            assert (cgh[f] == co[f]).all(), (f, lvl)
        assert no > 0
        cloud = orc.projectToPointCloud(lastDepth, K, lvl)
        for sigma in (float(np.sqrt(no)), -1.0):
            Ao, bo = orc.rgbStep(co, sigma, cloud, float(cam[0]), float(cam[1]), dx, dy, sobelScale)
            Ag, bg = dms.ops.rgbStep(co, sigma, cloud, float(cam[0]), float(cam[1]), dx, dy, sobelScale)
            _sum_close(Ag, Ao, what="rgb A level %d" % lvl)
            _sum_close(bg, bo, rtol=2e-3, what="rgb b level %d" % lvl)


def test_so3Step_parity(dms, orc, tracker_pair):
    g, o = tracker_pair
    lvl = 2
    K = [528.0 / 4, 528.0 / 4, 320.0 / 4, 240.0 / 4]
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float64)
    last, nxt = o.buffer(8, lvl), o.buffer(7, lvl)
    for ang in (0.0, 0.02):
        Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        ib = (Km @ Rm @ np.linalg.inv(Km)).astype(np.float32)
        ki = np.linalg.inv(Km).astype(np.float32)
        kr = (Km @ Rm).astype(np.float32)
        Ao, bo, ro = orc.so3Step(last, nxt, ib, ki, kr)
        Ag, bg, rg = dms.ops.so3Step(last, nxt, ib, ki, kr)
        assert rg[1] == ro[1]
        _sum_close(Ag, Ao, what="so3 A")
        _sum_close(bg, bo, rtol=2e-3, what="so3 b")
        _sum_close(rg[0], ro[0], what="so3 residual")


# ------------------------------------------------------------------------------------------
# whole tracker: device-resident Gauss-Newton vs the oracle's host loop
# ------------------------------------------------------------------------------------------
CONFIGS = {
    # BASELINE config 2: ICP only, --fo, single pyramid level  => iterations {3,0,0}
    "C2_icp_fast": dict(rgbOnly=False, icpWeight=100.0, pyramid=False, fastOdom=True, so3=False),
    # BASELINE config 3: full 3-level ICP + RGB with SO3 pre-alignment => {10,5,4}
    "C3_full": dict(rgbOnly=False, icpWeight=10.0, pyramid=True, fastOdom=False, so3=True),
    # GPUTest protocol (GPUTest.cpp:278): no pyramid, so3
    "gputest": dict(rgbOnly=False, icpWeight=10.0, pyramid=False, fastOdom=False, so3=True),
    "rgb_only": dict(rgbOnly=True, icpWeight=10.0, pyramid=True, fastOdom=False, so3=False),
    "icp_pyramid": dict(rgbOnly=False, icpWeight=100.0, pyramid=True, fastOdom=False, so3=False),
}


def _fresh_pair(dms, orc, gputest_pair):
    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    rgba1, rgba2 = helpers.rgba(gputest_pair["rgb1"]), helpers.rgba(gputest_pair["rgb2"])
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    pose = np.eye(4, dtype=np.float32)
    for trk in (g, o):
        trk.initICPModel(verts, norms, 20.0, pose)
        trk.initRGBModel(rgba1)
        trk.initICP(gputest_pair["depth2"], 20.0)
        trk.initRGB(rgba2)
        trk.initFirstRGB(rgba1)
    return g, o


@pytest.fixture
def track_mode(request, monkeypatch):
    """'persistent' = one resident kernel per pyramid level, grid-wide sums through integer atomics (default);
    'launches' = pass1 / pass2 / solve launches per iteration (fallback path).  Read by the library once, when a tracker is
    created.  Both compute the order-free sums of csrc/canon.hpp and the canonical scalar section: same bits."""
    # 'coarse' (round 6): SO3 + level 2 + level 1 in ONE resident launch (k_track_coarse, DMS_TRACK_FUSE=1) - built, the same bits,
    # slower on the MI355X than a launch per stage, so off by default and kept under test
    monkeypatch.delenv("DMS_TRACK_FUSE", raising=False)
    if request.param == "launches":
        monkeypatch.setenv("DMS_TRACK_MODE", "launches")
    else:
        monkeypatch.delenv("DMS_TRACK_MODE", raising=False)
        if request.param == "coarse":
            monkeypatch.setenv("DMS_TRACK_FUSE", "1")
    return request.param


def _assert_results_identical(rg, ro, g, what):
    assert list(rg.iterations_run) == list(ro.iterations_run), what
    assert rg.so3_iterations_run == ro.so3_iterations_run, what
    assert rg.rejected_jump == ro.rejected_jump, what
    for f in ("lastICPError", "lastICPCount", "lastRGBError", "lastRGBCount", "lastSO3Error", "lastSO3Count"):
        a, b = np.float32(getattr(rg, f)), np.float32(getattr(ro, f))
        assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (what, f, a, b)
    assert helpers.nan_equal(np.array(rg.lastA), np.array(ro.lastA)), what + " lastA"
    assert helpers.nan_equal(np.array(rg.lastb), np.array(ro.lastb)), what + " lastb"
    assert g.canonRetries() == ro.canon_retries, (what, g.canonRetries(), ro.canon_retries)


@pytest.mark.parametrize("track_mode", ["persistent", "launches", "coarse"], indirect=True)
@pytest.mark.parametrize("name", list(CONFIGS))
def test_track_pose_parity_gputest_pair(dms, orc, gputest_pair, name, track_mode):
    """Whole tracker calls on the reference's GPUTest pair: pose, side outputs, the 6x6 system and the iteration counts of
    the HIP path equal the oracle's BIT FOR BIT in both execution modes."""
    cfg = CONFIGS[name]
    g, o = _fresh_pair(dms, orc, gputest_pair)
    t0, R0 = np.zeros(3, np.float32), np.eye(3, dtype=np.float32)
    tg, Rg, rg = g.getIncrementalTransformation(t0, R0, **cfg)
    to, Ro, ro = o.getIncrementalTransformation(t0, R0, **cfg)
    helpers.assert_pose_identical(tg, Rg, to, Ro, what=name)
    _assert_results_identical(rg, ro, g, name)
    cov_g = g.getCovariance()
    cov_o = orc.covariance(np.array(ro.lastA))
    assert np.allclose(np.diag(cov_g), np.diag(cov_o), rtol=1e-9)
    # the motion between the two GPUTest frames is small but non-zero
    assert 1e-4 < np.linalg.norm(tg) < 0.1


@pytest.mark.parametrize("track_mode", ["persistent", "launches", "coarse"], indirect=True)
@pytest.mark.parametrize("name", ["C2_icp_fast", "C3_full", "gputest", "rgb_only"])
def test_track_reproduces_committed_golden_vectors(dms, gputest_pair, name, track_mode):
    """The committed golden vectors of the tracker on the reference's GPUTest pair
    (tests/golden/oracle_gputest.npz) against the HIP path alone — no oracle in this test: the same bits."""
    import os

    want = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_gputest.npz"))
    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    rgba1, rgba2 = helpers.rgba(gputest_pair["rgb1"]), helpers.rgba(gputest_pair["rgb2"])
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    g.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
    g.initRGBModel(rgba1)
    g.initICP(gputest_pair["depth2"], 20.0)
    g.initRGB(rgba2)
    g.initFirstRGB(rgba1)
    tg, Rg, rg = g.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **CONFIGS[name])
    # (the tracker object evaluates its rows with fused multiply-adds: the "_fma" vectors of the oracle)
    helpers.assert_pose_identical(tg, Rg, want[name + "_fma_t"], want[name + "_fma_R"], what=name)
    c = want[name + "_fma_counts"]  # ICP count, RGB count, SO3 count, SO3 iterations, iterations per level
    assert [rg.so3_iterations_run] + list(rg.iterations_run) == [int(v) for v in c[3:7]]
    for got, ref in ((rg.lastICPCount, c[0]), (rg.lastRGBCount, c[1]), (rg.lastSO3Count, c[2])):
        assert got == ref, (name, got, ref)
    assert np.array(rg.lastA).tobytes() == np.asarray(want[name + "_fma_lastA"], np.float64).tobytes(), name


@pytest.mark.parametrize("track_mode", ["persistent", "launches"], indirect=True)
@pytest.mark.parametrize("case", ["no_live_depth", "black_live_image", "no_depth_and_black"])
@pytest.mark.parametrize("early_exit", [False, True])
def test_track_degenerate_inputs(dms, orc, gputest_pair, track_mode, case, early_exit, monkeypatch):
    """Degenerate frames: no live depth at all (zero ICP correspondences: the 6x6 system is singular
    and the pivoted LDLT semantics apply) and a black live image (zero photometric correspondences:
    sigma of an empty set).  Every block of a resident kernel must take the same path; the result
    must equal the oracle's, NaN for NaN."""
    # early_exit: the resident-kernel instantiation that leaves a level after an iteration without any correspondence
    # (what the frame step uses for its model-to-model pass); the results must not depend on it
    monkeypatch.setenv("DMS_TRACK_EARLY_EXIT", "1" if early_exit else "0")
    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    rgba1, rgba2 = helpers.rgba(gputest_pair["rgb1"]), helpers.rgba(gputest_pair["rgb2"])
    depth2 = gputest_pair["depth2"].copy()
    if case in ("no_live_depth", "no_depth_and_black"):
        depth2[:] = 0
    if case in ("black_live_image", "no_depth_and_black"):  # (both: no correspondence of either kind — the resident kernels leave a level early)
        rgba2 = np.zeros_like(rgba2)
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    for trk in (g, o):
        trk.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        trk.initRGBModel(rgba1)
        trk.initICP(depth2, 20.0)
        trk.initRGB(rgba2)
        trk.initFirstRGB(rgba1)
    cfg = CONFIGS["C3_full"]
    t0, R0 = np.zeros(3, np.float32), np.eye(3, dtype=np.float32)
    tg, Rg, rg = g.getIncrementalTransformation(t0, R0, **cfg)
    to, Ro, ro = o.getIncrementalTransformation(t0, R0, **cfg)
    helpers.assert_pose_identical(tg, Rg, to, Ro, what=case)
    _assert_results_identical(rg, ro, g, case)
    if case in ("no_live_depth", "no_depth_and_black"):
        assert rg.lastICPCount == ro.lastICPCount == 0
    if case in ("black_live_image", "no_depth_and_black"):
        assert rg.lastRGBCount == ro.lastRGBCount == 0
    assert np.array_equal(np.isnan(np.array(rg.lastA)), np.isnan(np.array(ro.lastA)))


def test_track_from_nonidentity_prior_and_second_call(dms, orc, gputest_pair):
    """Prior pose with rotation + translation; two consecutive calls (the SO3 image swap of
    RGBDOdometry.cpp:595-601 changes what the second call sees)."""
    g, o = _fresh_pair(dms, orc, gputest_pair)
    ang = 0.2
    R0 = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    t0 = np.array([0.3, -0.1, 0.5], np.float32)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3], pose[:3, 3] = R0, t0
    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    for trk in (g, o):
        trk.initICPModel(verts, norms, 20.0, pose)
    cfg = CONFIGS["C3_full"]
    for call in range(2):
        tg, Rg, rg = g.getIncrementalTransformation(t0, R0, **cfg)
        to, Ro, ro = o.getIncrementalTransformation(t0, R0, **cfg)
        helpers.assert_pose_identical(tg, Rg, to, Ro, what="call %d" % call)
        _assert_results_identical(rg, ro, g, "call %d" % call)
        for lvl in range(3):
            assert (g.buffer(7, lvl) == o.buffer(7, lvl)).all()  # nextImage after the swap
            assert (g.buffer(8, lvl) == o.buffer(8, lvl)).all()  # lastNextImage after the swap


def test_track_recovers_known_motion_synthetic(dms, orc):
    """Size-independent property: on a noise-free synthetic pair with known camera motion the
    device-resident tracker recovers the motion (and agrees with the oracle to the bar)."""
    from densemonoslam_amd import synth

    K = synth.K_640
    d1, rgb1, T1 = synth.frame(10, noise=False)
    d2, rgb2, T2 = synth.frame(11, noise=False)
    # model maps = frame 1 geometry in its own camera frame (copyMaps layout), pose T1
    vo = orc.createVMap(K, d1, 20.0)
    no = orc.createNMap(vo)
    H = 480
    verts = np.zeros((480, 640, 4), np.float32)
    norms = np.zeros((480, 640, 4), np.float32)
    ok = ~np.isnan(vo[:H]) & ~np.isnan(no[:H])
    for c in range(3):
        verts[..., c] = np.where(ok, vo[c * H:(c + 1) * H], 0)
        norms[..., c] = np.where(ok, no[c * H:(c + 1) * H], 0)
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    P1 = T1.astype(np.float32)
    for trk in (g, o):
        trk.initICPModel(verts, norms, 20.0, P1)
        trk.initRGBModel(synth.rgba(rgb1))
        trk.initICP(d2, 20.0)
        trk.initRGB(synth.rgba(rgb2))
        trk.initFirstRGB(synth.rgba(rgb1))
    # full configuration: the box room is near-planar, so ICP alone slides along the wall; the
    # photometric term (checker texture) and the SO3 pre-alignment pin the motion down
    cfg = dict(rgbOnly=False, icpWeight=10.0, pyramid=True, fastOdom=False, so3=True)
    tg, Rg, rg = g.getIncrementalTransformation(P1[:3, 3], P1[:3, :3], **cfg)
    to, Ro, ro = o.getIncrementalTransformation(P1[:3, 3], P1[:3, :3], **cfg)
    helpers.assert_pose_identical(tg, Rg, to, Ro, what="synthetic vs oracle")
    _assert_results_identical(rg, ro, g, "synthetic")
    assert rg.lastRGBCount > 10000 and rg.lastICPCount > 200000
    # against ground truth: the estimate must be much closer to the true pose than the prior was
    err_t = np.linalg.norm(tg.astype(np.float64) - T2[:3, 3])
    err_r = helpers.rot_angle_deg(Rg, T2[:3, :3])
    prior_t = np.linalg.norm(T1[:3, 3] - T2[:3, 3])
    prior_r = helpers.rot_angle_deg(T1[:3, :3], T2[:3, :3])
    assert err_t < 0.35 * prior_t, (err_t, prior_t)
    assert err_r < 0.75 * prior_r, (err_r, prior_r)


@pytest.mark.parametrize("track_mode", ["persistent", "launches"], indirect=True)
def test_track_profiling_counters(dms, orc, gputest_pair, track_mode):
    g, _ = _fresh_pair(dms, orc, gputest_pair)
    g.set_profiling(True)
    _, _, r = g.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **CONFIGS["C3_full"])
    assert list(r.iterations_run) == [10, 5, 4]
    if track_mode == "persistent":
        for lvl in range(3):  # one resident kernel per pyramid level
            ms, n = g.kernel_time("gn_level%d" % lvl)
            assert n == 1 and ms > 0
        assert g.kernel_time("gn_pass1")[1] == 0
        # in-kernel phase clocks of block 0 (pass 1, ..., solve) are exposed as "phase:<i>"
        assert g.kernel_time("phase:1")[0] > 0 and g.kernel_time("phase:39")[0] > 0  # level * 16 + phase
    else:
        ms1, n1 = g.kernel_time("gn_pass1")
        ms2, n2 = g.kernel_time("gn_pass2")
        ms3, n3 = g.kernel_time("gn_solve")
        assert n1 == 19 and n2 == 19 and n3 == 19
        assert ms1 > 0 and ms2 > 0 and ms3 > 0
    if track_mode == "persistent":
        assert g.kernel_time("so3_level")[1] == 1 and g.kernel_time("so3_pass")[1] == 0
    else:
        assert g.kernel_time("so3_pass")[1] == 10


def test_error_paths(dms):
    from densemonoslam_amd import capi
    import ctypes as C

    # workspace too small -> DMS_ERR_WORKSPACE, message set, no crash
    small = capi.DeviceBuffer(64)
    img = capi.DeviceImage(3 * 16, 16, np.float32)
    R, t, k = capi.mat33(np.eye(3)), capi.float3(np.zeros(3)), capi.Camera(100, 100, 8, 8)
    A = (C.c_float * 36)()
    b = (C.c_float * 6)()
    r = (C.c_float * 2)()
    rc = capi.lib.dms_icpStep(C.byref(R), C.byref(t), img.ref, img.ref, C.byref(R), C.byref(t), C.byref(k), img.ref, img.ref, 0.1, 0.3,
                              C.c_void_p(small.ptr), 64, A, b, r, 0, 0, None)
    assert rc == -3 and b"workspace" in capi.lib.dms_last_error()
    # shape mismatch -> DMS_ERR_INVALID_ARG
    src = capi.DeviceImage(32, 32, np.uint16)
    dst = capi.DeviceImage(10, 10, np.uint16)
    assert capi.lib.dms_pyrDown(src.ref, dst.ref, None) == -1
    # null pointers
    assert capi.lib.dms_createNMap(None, None, None) == -1


@pytest.mark.parametrize("track_mode", ["persistent", "launches"], indirect=True)
@pytest.mark.parametrize("bias", [-12, -30])
def test_exponents_too_small_repeat_the_reduction_on_both_sides(dms, orc, gputest_pair, track_mode, bias):
    """Canonical sums (csrc/canon.hpp): a reduction whose diagonal totals do not fit the grid its column exponents promised
    is repeated with every exponent raised by 8 — by the whole resident grid on a word set of the launch's pool, by the
    solving block alone in `launches` mode, by a loop in the oracle.  Forced here through the static exponents of the
    call's first reductions: the same number of repetitions and the same bits as the oracle under the same bias."""
    g, o = _fresh_pair(dms, orc, gputest_pair)
    g.setExpBias(bias)
    o.setExpBias(bias)
    cfg = CONFIGS["C3_full"]
    t0, R0 = np.zeros(3, np.float32), np.eye(3, dtype=np.float32)
    tg, Rg, rg = g.getIncrementalTransformation(t0, R0, **cfg)
    to, Ro, ro = o.getIncrementalTransformation(t0, R0, **cfg)
    assert ro.canon_retries >= 2  # (the SO3 stage's first reduction and the first Gauss-Newton one, at least)
    helpers.assert_pose_identical(tg, Rg, to, Ro, what="bias %d" % bias)
    _assert_results_identical(rg, ro, g, "bias %d" % bias)


def test_pose_does_not_depend_on_the_grid_size(dms, gputest_pair, monkeypatch):
    """Integer sums are order free: resident kernels on differently sized grids (DMS_PERSIST_BLOCKS moves the pixels-per-thread
    choice of the small levels) and the launch-per-phase kernels give the same bits."""
    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    rgba1, rgba2 = helpers.rgba(gputest_pair["rgb1"]), helpers.rgba(gputest_pair["rgb2"])
    outs = []
    for blocks, mode, cap in ((160, None, 0), (40, None, 0), (96, None, 0), (160, "launches", 0), (160, None, 120), (160, None, 20)):
        # cap: a device with that many compute units — 120: level 0 takes four pixels per thread; 20: no level fits, every level
        # runs launch-per-phase while the SO3 stage (38 blocks) does too
        monkeypatch.setenv("DMS_PERSIST_BLOCKS", str(blocks))
        if cap:
            monkeypatch.setenv("DMS_PERSIST_MAX_BLOCKS", str(cap))
        else:
            monkeypatch.delenv("DMS_PERSIST_MAX_BLOCKS", raising=False)
        if mode:
            monkeypatch.setenv("DMS_TRACK_MODE", mode)
        else:
            monkeypatch.delenv("DMS_TRACK_MODE", raising=False)
        g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
        if cap:
            assert g.getMode()[1] == cap
        g.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        g.initRGBModel(rgba1)
        g.initICP(gputest_pair["depth2"], 20.0)
        g.initRGB(rgba2)
        g.initFirstRGB(rgba1)
        t, R, r = g.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **CONFIGS["C3_full"])
        outs.append(t.tobytes() + R.tobytes() + np.array(r.lastA).tobytes() + np.array(r.lastb).tobytes())
        g.close()
    # the per-handle form of the cap (dms_odometry_set_resident_budget: what dms_session gives cameras that share the device), set and
    # lifted on one handle between calls
    from densemonoslam_amd.capi import lib

    monkeypatch.delenv("DMS_PERSIST_MAX_BLOCKS", raising=False)
    monkeypatch.setenv("DMS_PERSIST_BLOCKS", "160")
    g = dms.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
    for cap, unchained in ((120, 1), (0, 0), (20, 0)):
        assert lib.dms_odometry_set_resident_budget(g.h, cap, unchained) == 0
        g.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        g.initRGBModel(rgba1)
        g.initICP(gputest_pair["depth2"], 20.0)
        g.initRGB(rgba2)
        g.initFirstRGB(rgba1)
        t, R, r = g.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **CONFIGS["C3_full"])
        outs.append(t.tobytes() + R.tobytes() + np.array(r.lastA).tobytes() + np.array(r.lastb).tobytes())
    g.close()
    assert all(o == outs[0] for o in outs[1:])


def test_resident_timeout_repeats_the_call_launch_per_phase(dms, orc, gputest_pair):
    """A resident kernel that times out at a grid-wide wait (injected) means its blocks were not all on the device.  The
    synchronous call repeats itself launch-per-phase at once — the result equals the oracle's bit for bit — and the handle
    stays in that mode."""
    from densemonoslam_amd.capi import lib

    import os

    if os.environ.get("DMS_TRACK_MODE") == "launches":
        pytest.skip("the environment already selects the launch-per-phase tracker")
    g, o = _fresh_pair(dms, orc, gputest_pair)
    assert g.getMode()[0] and not g.getMode()[2]
    assert lib.dms_odometry_inject_timeout(g.h, 1) == 0
    cfg = CONFIGS["C3_full"]
    t0, R0 = np.zeros(3, np.float32), np.eye(3, dtype=np.float32)
    tg, Rg, rg = g.getIncrementalTransformation(t0, R0, **cfg)
    to, Ro, ro = o.getIncrementalTransformation(t0, R0, **cfg)
    helpers.assert_pose_identical(tg, Rg, to, Ro, what="repeated call")
    _assert_results_identical(rg, ro, g, "repeated call")
    resident, _, fell_back = g.getMode()
    assert not resident and fell_back
    for lvl in range(3):  # the SO3 image swap happened exactly once
        assert (g.buffer(7, lvl) == o.buffer(7, lvl)).all() and (g.buffer(8, lvl) == o.buffer(8, lvl)).all()
    tg, Rg, rg = g.getIncrementalTransformation(t0, R0, **cfg)
    to, Ro, ro = o.getIncrementalTransformation(t0, R0, **cfg)
    helpers.assert_pose_identical(tg, Rg, to, Ro, what="next call")
