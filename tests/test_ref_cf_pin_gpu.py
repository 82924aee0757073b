"""The HIP path's pyramid / preparation operators, its FUSED pyramid kernels and its NID scores against what the REFERENCE's own
Cuda/cudafuncs.cu returned (tests/golden/ref_cudafuncs.npz, see tests/test_ref_cf_pin_cpu.py), and against the reference run
live in the same process on other inputs (oracle/_ref/libref_cudafuncs.so, skipped when it was not built): product-vs-reference
for SURVEY 8 a7 / f3.  The last test feeds the tracking steps' reference pin (tests/test_ref_live_gpu.py) with pyramids the
REFERENCE built, so that a wrong pyramid cannot hide behind identical inputs on both sides."""
import os

import numpy as np
import pytest

from tests import helpers, ref_cases, ref_cases_cf as cf
from tests.test_ref_cf_pin_cpu import GOLDEN, load_case

pytestmark = pytest.mark.gpu
NID_TOL = 3e-5


def _check(out, fx, fed, what):
    kinds = {}
    for name, got in out.items():
        if name == "nid":
            continue
        kinds[name] = cf.compare(name, got, fx, cf.tol_of(name, fed))
        if cf.tol_of(name, fed) == 0.0:
            assert kinds[name] == "exact", (what, name)
    # NID: the product's histograms are the reference's counts; its entropy sums are fp64 tree sums where the reference's host loop
    # accumulates double terms into a FLOAT variable one by one (cudafuncs.cu:1589-1603; restated exactly by oracle/orc_nid.c, which
    # the CPU test holds to these values bit for bit): a sequential rounding chain a device has no reason to reproduce.  NID_TOL covers
    # that accumulation noise (observed <= 1e-5 on a score in [0, 1] that is compared with a threshold of 0.8).
    assert np.abs(out["nid"].astype(np.float64) - np.asarray(fx["nid"], np.float64)).max() <= NID_TOL, (what, out["nid"], fx["nid"])
    return kinds


@pytest.mark.parametrize("case", ["small", "full"])
def test_product_operators_equal_the_references(orc, case):
    from densemonoslam_amd import capi, odometry

    assert capi.device_count() >= 1, "no MI355X visible"
    fx = load_case(case)
    pair = np.load(os.path.join(os.path.dirname(GOLDEN), "gputest_pair.npz"))
    inp = cf.inputs(case, pair, orc)
    for k, v in cf.input_hashes(inp).items():
        assert str(v) == str(fx[k]), k
    fed = case == "small"
    kinds = _check(cf.chain(cf.HipOps(odometry.ops), inp, feed=fx if fed else None), fx, fed, case)
    assert len(kinds) >= 40


@pytest.mark.parametrize("case", ["small", "full"])
def test_fused_pyramid_kernels_equal_the_references(orc, case):
    """RGBDOdometry::init* of the product (k_live_levels, k_model_levels012 and the operator chain behind initRGB*) leaves in the
    tracker's pyramid buffers what the reference's operator sequence produced."""
    from densemonoslam_amd import odometry

    fx = load_case(case)
    pair = np.load(os.path.join(os.path.dirname(GOLDEN), "gputest_pair.npz"))
    inp = cf.inputs(case, pair, orc)
    K = [float(v) for v in inp["K"]]
    H, W = inp["depth"].shape
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3], pose[:3, 3] = cf.POSE_R, cf.POSE_T
    rgb1, rgb2 = np.asarray(pair["rgb1"]), np.asarray(pair["rgb2"])
    if case == "small":
        rgb1, rgb2 = (np.ascontiguousarray(a[2::5, 1::5][cf.CROP]) for a in (rgb1, rgb2))
    g = odometry.RGBDOdometry(W, H, K[2], K[3], K[0], K[1])
    g.initICPModel(inp["verts4"], inp["norms4"], cf.CUTOFF, pose)
    g.initRGBModel(helpers.rgba(rgb1))
    g.initICP(inp["depth"], cf.CUTOFF)
    g.initFirstRGB(helpers.rgba(rgb1))
    names = {0: "vmap_L%d", 1: "nmap_L%d", 2: "tvmap_L%d", 3: "tnmap_L%d", 4: "dmap_L%d", 6: "img1_L%d", 8: "img1_L%d"}
    n = 0
    for which, pat in names.items():
        for lvl in range(3):
            name = pat % lvl
            if name in ("img1_L0", ):
                continue  # imageBGRToIntensity's output: the one operator without a reference (an input of the chain)
            got = cf.canon(g.buffer(which, lvl), planes3=which in (0, 1, 2, 3))
            tol = cf.tol_of(name, False)
            kind = cf.compare(name, got, fx, tol)
            assert tol > 0 or kind == "exact", (name, kind)
            n += 1
    # the live colour side: initRGB after an initICPMaps of the same vertices (vmaps_tmp is what populateRGBDData reads, :209-213)
    g.initICPMaps(inp["verts4"], inp["norms4"], cf.CUTOFF)
    g.initRGB(helpers.rgba(rgb2))
    for lvl in (1, 2):
        assert cf.compare("img2_L%d" % lvl, cf.canon(g.buffer(7, lvl)), fx) == "exact"
        assert cf.compare("dmap_L%d" % lvl, cf.canon(g.buffer(5, lvl)), fx) == "exact"
    g.close()
    assert n >= 19
    # the frame step's model half in one launch (k_model_levels012 behind dms_odometry_initModelFused): same buffers, same bits
    g2 = odometry.RGBDOdometry(W, H, K[2], K[3], K[0], K[1])
    g2.initModelFused(inp["verts4"], inp["norms4"], helpers.rgba(rgb1), inp["verts4_old"], inp["norms4"], helpers.rgba(rgb2), False, False, pose)
    for which, pat in ((2, "tvmap_L%d"), (3, "tnmap_L%d"), (4, "dmap_L%d"), (6, "img1_L%d")):
        for lvl in range(3):
            name = pat % lvl
            if name == "img1_L0":
                continue
            tol = cf.tol_of(name, False)
            kind = cf.compare(name, cf.canon(g2.buffer(which, lvl), planes3=which in (2, 3)), fx, tol)
            assert tol > 0 or kind == "exact", ("fused", name, kind)
    g2.close()


def _synth_inputs(orc, synth, W, H, K, k0):
    d1, rgb1, _ = synth.frame(k0, width=W, height=H, K=K, noise=True)
    d2, rgb2, _ = synth.frame(k0 + 1, width=W, height=H, K=K, noise=True)
    vo = orc.createVMap(K, d1, 20.0)
    no = orc.createNMap(vo)
    verts = np.zeros((H, W, 4), np.float32)
    norms = np.zeros((H, W, 4), np.float32)
    ok = ~np.isnan(vo[:H]) & ~np.isnan(no[:H])
    for c in range(3):
        verts[..., c] = np.where(ok, vo[c * H:(c + 1) * H], 0)
        norms[..., c] = np.where(ok, no[c * H:(c + 1) * H], 0)
    return dict(K=np.array(K, np.float64), depth=d2, verts4=verts, norms4=norms, verts4_old=np.roll(verts, 5, axis=0),
                img1=orc.imageBGRToIntensity(synth.rgba(rgb1)), img2=orc.imageBGRToIntensity(synth.rgba(rgb2)))


@pytest.mark.parametrize("size", [(320, 240, 3), (333, 251, 40), (1241, 376, 7)])
def test_product_operators_equal_the_references_live(orc, size):
    from densemonoslam_amd import odometry, synth
    from oracle import ref_cf

    if not ref_cf.available():
        pytest.skip("oracle/_ref/libref_cudafuncs.so not built (needs /root/reference at build time)")
    W, H, k0 = size
    K = (264.0 * W / 320.0, 264.0 * W / 320.0, W / 2.0, H / 2.0)
    inp = _synth_inputs(orc, synth, W, H, K, k0)
    theirs = cf.chain(ref_cf, inp)
    ours = cf.chain(cf.HipOps(odometry.ops), inp)
    mine = cf.chain(cf.OrcOps(orc), inp)
    for name in theirs:
        for what, got in (("product", ours[name]), ("restatement", mine[name])):
            if name == "nid":
                assert np.abs(got.astype(np.float64) - theirs[name]).max() <= (NID_TOL if what == "product" else 0.0), (what, got, theirs[name])
                continue
            tol = cf.tol_of(name, False)
            a, b = theirs[name], got
            if a.tobytes() == b.tobytes():
                continue
            assert tol > 0, (what, name, int((a != b).sum()))
            assert (np.isnan(a) == np.isnan(b)).all(), (what, name)
            assert np.nan_to_num(np.abs(a.astype(np.float64) - b)).max() <= tol, (what, name)


def test_tracking_steps_on_reference_built_pyramids(orc):
    """The four tracking steps, product vs the reference's reduce.cu live, on pyramids built by the REFERENCE's cudafuncs.cu from
    the GPUTest pair (not by the restatement): inputs and steps are then both the reference's."""
    from densemonoslam_amd import odometry
    from oracle import ref, ref_cf
    from tests.test_ref_pin_gpu import _Ops

    if not (ref.available() and ref_cf.available()):
        pytest.skip("oracle/_ref libraries not built (need /root/reference at build time)")
    pair = np.load(os.path.join(os.path.dirname(GOLDEN), "gputest_pair.npz"))
    inp = cf.inputs("full", pair, orc)
    c = cf.chain(ref_cf, inp)
    ours, theirs = _Ops(odometry.ops), ref
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    K = [float(v) for v in inp["K"]]
    n_corr = 0
    for lvl in range(3):
        cam = np.array([np.float32(v) / np.float32(1 << lvl) for v in K], np.float32)
        Km = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float64)
        # model maps as the reference built them (identity model pose: the untransformed maps), live maps from the live depth
        vm, nm, vc, nc = c["mvmap_L%d" % lvl], c["mnmap_L%d" % lvl], c["vmap_L%d" % lvl], c["nmap_L%d" % lvl]
        for R, t in ref_cases.POSES[:2]:
            a = [R.astype(np.float32), t.astype(np.float32), vc, nc, I3, z3, cam, vm, nm, ref_cases.DIST_THRES, ref_cases.ANGLE_THRES]
            Ao, bo, ro = ours.icpStep(*a)
            Ar, br, rr = theirs.icpStep(*a)
            assert ro[1] == rr[1] and rr[1] > 100, ("inliers", lvl, ro[1], rr[1])
            assert np.abs(Ao.astype(np.float64) - Ar).max() <= 2e-6 * max(np.abs(Ar).max(), 1e-30)
            Ri = np.linalg.inv(np.vstack([np.hstack([R, t.reshape(3, 1)]), [0, 0, 0, 1]]))
            krkinv = (Km @ Ri[:3, :3] @ np.linalg.inv(Km)).astype(np.float32)
            kt = (Km @ Ri[:3, 3]).astype(np.float32)
            img1 = inp["img1"] if lvl == 0 else c["img1_L%d" % lvl]
            img2 = inp["img2"] if lvl == 0 else c["img2_L%d" % lvl]
            # `last` = frame 1 (model depth pyramid), `next` = frame 2: its depth pyramid stands in with the model's (the photometric
            # step only gates on it), gradients and point cloud from the reference's own operators
            b_ = [float(ref_cases.MIN_GRAD[lvl] ** 2 / ref_cases.SOBEL_SCALE ** 2), c["dIdx_L%d" % lvl], c["dIdy_L%d" % lvl], c["dmap_L%d" % lvl],
                  c["dmap_L%d" % lvl], img1, img2, ref_cases.MAX_DEPTH_DELTA, kt, krkinv]
            co, so, no_ = ours.computeRgbResidual(*b_)
            cr, sr, nr = theirs.computeRgbResidual(*b_)
            assert (so, no_) == (sr, nr), lvl
            v = cr["valid"] != 0
            assert ((co["valid"] != 0) == v).all()
            for f in ("zero_x", "zero_y", "one_x", "one_y", "diff"):
                assert (co[f][v] == cr[f][v]).all(), (f, lvl)
            n_corr += int(nr)
            c_ = [cr, float(np.sqrt(max(nr, 1))), c["cloud_L%d" % lvl], float(cam[0]), float(cam[1]), c["dIdx_L%d" % lvl], c["dIdy_L%d" % lvl],
                  ref_cases.SOBEL_SCALE]
            Ao, bo = ours.rgbStep(*c_)
            Ar, br = theirs.rgbStep(*c_)
            assert np.abs(Ao.astype(np.float64) - Ar).max() <= 2e-6 * max(np.abs(Ar).max(), 1e-30), ("rgb A", lvl)
    assert n_corr > 1000
