"""The HIP path's tracking steps against the REFERENCE's own kernels run LIVE in the same process (oracle/_ref/libref_reduce.so,
built by oracle/ref_build.sh where /root/reference exists and carried to the GPU box with the snapshot) on inputs other than
the committed fixture's: synthetic frame pairs at 320 x 240 and 333 x 251.  Skipped when the library is not there."""
import numpy as np
import pytest

from tests import ref_cases

pytestmark = pytest.mark.gpu


def _inputs(orc, synth, W, H, K, k0):
    from tests import helpers  # noqa: F401

    d1, rgb1, T1 = synth.frame(k0, width=W, height=H, K=K, noise=True)
    d2, rgb2, T2 = synth.frame(k0 + 1, width=W, height=H, K=K, noise=True)
    vo = orc.createVMap(K, d1, 20.0)
    no = orc.createNMap(vo)
    verts = np.zeros((H, W, 4), np.float32)
    norms = np.zeros((H, W, 4), np.float32)
    ok = ~np.isnan(vo[:H]) & ~np.isnan(no[:H])
    for c in range(3):
        verts[..., c] = np.where(ok, vo[c * H:(c + 1) * H], 0)
        norms[..., c] = np.where(ok, no[c * H:(c + 1) * H], 0)
    o = orc.Odometry(W, H, K[2], K[3], K[0], K[1])
    o.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
    o.initRGBModel(synth.rgba(rgb1))
    o.initICP(d2, 20.0)
    o.initRGB(synth.rgba(rgb2))
    o.initFirstRGB(synth.rgba(rgb1))
    rel = np.linalg.inv(T1) @ T2  # the true relative motion: realistic correspondences
    lv = []
    for lvl in range(3):
        d = dict(vmap_curr=o.buffer(0, lvl), nmap_curr=o.buffer(1, lvl), vmap_g_prev=o.buffer(2, lvl), nmap_g_prev=o.buffer(3, lvl),
                 lastDepth=o.buffer(4, lvl), nextDepth=o.buffer(5, lvl), lastImage=o.buffer(6, lvl), nextImage=o.buffer(7, lvl),
                 lastNextImage=o.buffer(8, lvl))
        d["dIdx"], d["dIdy"] = orc.computeDerivativeImages(d["nextImage"])
        d["cloud"] = orc.projectToPointCloud(d["lastDepth"], K, lvl)
        lv.append(d)
    return lv, rel


@pytest.mark.parametrize("size", [(320, 240, 3), (333, 251, 40)])
def test_steps_equal_the_references_live(orc, size):
    from densemonoslam_amd import capi, odometry, synth
    from oracle import ref
    from tests.test_ref_pin_gpu import _Ops

    if not ref.available():
        pytest.skip("oracle/_ref/libref_reduce.so not built (needs /root/reference at build time)")
    assert capi.device_count() >= 1
    W, H, k0 = size
    K = (264.0 * W / 320.0, 264.0 * W / 320.0, W / 2.0, H / 2.0)
    lv, rel = _inputs(orc, synth, W, H, K, k0)
    ours, theirs = _Ops(odometry.ops), ref
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    n_corr = 0
    for lvl, d in enumerate(lv):
        cam = np.array([np.float32(v) / np.float32(1 << lvl) for v in K], np.float32)
        Km = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float64)
        for R, t in ((np.eye(3), np.zeros(3)), (rel[:3, :3], rel[:3, 3])):
            a = [R.astype(np.float32), t.astype(np.float32), d["vmap_curr"], d["nmap_curr"], I3, z3, cam, d["vmap_g_prev"], d["nmap_g_prev"],
                 ref_cases.DIST_THRES, ref_cases.ANGLE_THRES]
            Ao, bo, ro = ours.icpStep(*a)
            Ar, br, rr = theirs.icpStep(*a)
            assert ro[1] == rr[1], ("inliers", lvl)
            scale = max(np.abs(Ar).max(), 1e-30)
            assert np.abs(Ao.astype(np.float64) - Ar).max() <= 2e-6 * scale and np.abs(bo.astype(np.float64) - br).max() <= 2e-6 * max(np.abs(br).max(), np.abs(Ar).max() ** 0.5, 1e-30)
            Ri = np.linalg.inv(np.vstack([np.hstack([R, t.reshape(3, 1)]), [0, 0, 0, 1]]))  # `next` seen from `last`
            krkinv = (Km @ Ri[:3, :3] @ np.linalg.inv(Km)).astype(np.float32)
            kt = (Km @ Ri[:3, 3]).astype(np.float32)
            b_ = [float(ref_cases.MIN_GRAD[lvl] ** 2 / ref_cases.SOBEL_SCALE ** 2), d["dIdx"], d["dIdy"], d["lastDepth"], d["nextDepth"], d["lastImage"],
                  d["nextImage"], ref_cases.MAX_DEPTH_DELTA, kt, krkinv]
            co, so, no_ = ours.computeRgbResidual(*b_)
            cr, sr, nr = theirs.computeRgbResidual(*b_)
            assert (so, no_) == (sr, nr), lvl
            v = cr["valid"] != 0
            assert ((co["valid"] != 0) == v).all()
            for f in ("zero_x", "zero_y", "one_x", "one_y", "diff"):
                assert (co[f][v] == cr[f][v]).all(), (f, lvl)
            n_corr += int(nr)
            for sg in (float(np.sqrt(max(nr, 1))), -1.0):
                c_ = [cr, sg, d["cloud"], float(cam[0]), float(cam[1]), d["dIdx"], d["dIdy"], ref_cases.SOBEL_SCALE]
                Ao, bo = ours.rgbStep(*c_)
                Ar, br = theirs.rgbStep(*c_)
                assert np.abs(Ao.astype(np.float64) - Ar).max() <= 2e-6 * max(np.abs(Ar).max(), 1e-30), ("rgb A", lvl)
            if lvl == 2:
                s_ = [d["lastNextImage"], d["nextImage"], (Km @ R @ np.linalg.inv(Km)).astype(np.float32), np.linalg.inv(Km).astype(np.float32),
                      (Km @ R).astype(np.float32)]
                Ao, bo, ro = ours.so3Step(*s_)
                Ar, br, rr = theirs.so3Step(*s_)
                assert ro[1] == rr[1]
                assert np.abs(Ao.astype(np.float64) - Ar).max() <= 2e-6 * max(np.abs(Ar).max(), 1e-30)
    assert n_corr > 1000


@pytest.mark.parametrize("pair", ["room_0_3", "room_40_44", "corner_10_12"])
def test_product_tracker_against_the_reference_driven_tracker_on_synthetic_pairs(orc, pair):
    """The whole-call evidence of tests/test_ref_pin_gpu.py (HIP tracker object against tracker calls whose every step of every
    iteration ran the REFERENCE's kernels) rests on ONE image pair, the GPUTest PNGs.  Here the same comparison runs LIVE on three more
    pairs - frames of the synthetic streams, 640 x 480, with motions of 1 - 4 cm / 0.3 - 1 degree between the two frames - for all five
    configurations: the north-star bar (1 mm, 0.01 degree) and the same iteration counts and breaks."""
    from densemonoslam_amd import capi, odometry, synth
    from oracle import ref
    from tests import helpers

    if not ref.available():
        pytest.skip("oracle/_ref/libref_reduce.so not built (needs /root/reference at build time)")
    assert capi.device_count() >= 1
    scene, k1, k2 = {"room_0_3": (None, 0, 3), "room_40_44": (None, 40, 44), "corner_10_12": (synth.CORNER_SCENE, 10, 12)}[pair]
    K = ref_cases.K
    d1, rgb1, _ = synth.frame(k1, width=640, height=480, K=K, noise=True, scene=scene)
    d2, rgb2, _ = synth.frame(k2, width=640, height=480, K=K, noise=True, scene=scene)
    # the GPUTest protocol's inputs (GPUTest.cpp:51-57,69-129): the first frame's depth on the TUM scale (1 / 5000 m), the second in mm
    p = {"rgb1": rgb1, "rgb2": rgb2, "depth1_raw": (d1.astype(np.uint32) * 5).clip(0, 65535).astype(np.uint16), "depth2": d2}
    want = ref_cases.run_trackers(orc, p, hooks=ref.step_hooks())
    verts, norms = helpers.gputest_model_maps(p["depth1_raw"], K)
    worst = (0.0, 0.0)
    for name, cfg in ref_cases.TRACKER_CONFIGS.items():
        g = odometry.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
        g.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        g.initRGBModel(helpers.rgba(rgb1))
        g.initICP(d2, 20.0)
        g.initRGB(helpers.rgba(rgb2))
        g.initFirstRGB(helpers.rgba(rgb1))
        t, R, res = g.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **cfg)
        dt, da = helpers.assert_pose_close(t, R, want["trk_%s_t" % name], want["trk_%s_R" % name], tol_m=1e-3, tol_deg=1e-2, what="%s %s" % (pair, name))
        worst = (max(worst[0], dt), max(worst[1], da))
        assert [res.so3_iterations_run] + list(res.iterations_run) == list(want["trk_%s_iters" % name]), (pair, name)
        g.close()
    moved = float(np.linalg.norm(want["trk_C3_full_t"]))
    assert moved > 2e-3, "the pair must carry a real motion (%.1e m)" % moved
    print("%s: worst difference to the reference-driven tracker %.2e m, %.2e deg (motion %.3f m)" % (pair, worst[0], worst[1], moved))
