"""CPU tests of the oracle (no GPU): golden vectors, analytic properties and the host-side math.

The reference holds no numeric test vectors for this path ("parity unpinned"), so
the oracle is anchored three ways: (1) committed golden vectors of the oracle on the reference's
own input fixture (regression pin), (2) properties that must hold for ANY correct restatement
(finite-difference Jacobians, conservation, idempotence, brute-force nearest surfel),
(3) independent numpy evaluations of individual formulas.
"""
import importlib.util
import os

import numpy as np
import pytest

from tests import helpers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_maker():
    spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(GOLDEN, "make_oracle_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_reproduces_golden_vectors():
    z = np.load(os.path.join(GOLDEN, "gputest_pair.npz"))
    want = np.load(os.path.join(GOLDEN, "oracle_gputest.npz"))
    got = _load_maker().compute(z)
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        if k.startswith(("icp_A", "icp_b")):
            # operator layer (orc_icpStep): fp64 accumulation, the OpenMP partial sums may be combined in a different order on
            # another core count
            assert np.allclose(a, b, rtol=1e-6, atol=1e-9), k
        else:
            # tracker object: order-free sums (orc_canon.c) + canonical scalar section (orc_scalar.c) — the same bits on any host
            assert a.shape == b.shape and a.tobytes() == np.asarray(b, a.dtype).tobytes(), k


def test_oracle_reproduces_fusion_golden(orc):
    """Fusion half of the frame step with the pose prior taken as is: every stored value comes from
    per-element arithmetic only, so the oracle must reproduce it bit for bit on any host."""
    spec = importlib.util.spec_from_file_location("make_fusion_golden", os.path.join(GOLDEN, "make_fusion_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = np.load(os.path.join(GOLDEN, "oracle_fusion.npz"))
    got = mod.compute()
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        a, b = np.asarray(got[k]), want[k]
        assert a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes(), k


def test_gputest_motion_is_small_and_consistent(orc):
    want = np.load(os.path.join(GOLDEN, "oracle_gputest.npz"))
    for name in ("C2_icp_fast", "C3_full", "gputest"):
        t, R = want[name + "_t"], want[name + "_R"]
        assert 1e-3 < np.linalg.norm(t) < 0.05
        assert helpers.rot_angle_deg(R, np.eye(3)) < 3.0
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)


def test_icp_row_is_the_point_to_plane_jacobian(orc):
    """Finite differences: row[0:6] = d residual / d (translation, rotation) of the twist applied in the
    previous-camera frame; residual = n . (s - d)."""
    rng = np.random.default_rng(0)
    H, W = 12, 16
    K = (20.0, 20.0, 8.0, 6.0)
    depth = (1500 + 200 * rng.random((H, W))).astype(np.uint16)
    vm = orc.createVMap(K, depth, 20.0)
    nm = orc.createNMap(vm)
    import ctypes as C

    def row_at(R, t, x, y):
        row = np.zeros(7, np.float32)
        args = [np.ascontiguousarray(R, np.float32).reshape(9), np.ascontiguousarray(t, np.float32), vm, nm, np.eye(3, dtype=np.float32).reshape(9),
                np.zeros(3, np.float32), vm, nm]
        p = [a.ctypes.data_as(C.c_void_p) for a in args]
        f = orc.lib.orc_icp_row(p[0], p[1], p[2], p[3], p[4], p[5], K[0], K[1], K[2], K[3], p[6], p[7], 0.5, 0.9, H, W, x, y, row.ctypes.data_as(C.c_void_p))
        return f, row

    x, y = 7, 5
    t0 = np.array([0.004, -0.003, 0.002], np.float32)
    f, row = row_at(np.eye(3), t0, x, y)
    assert f == 1
    # moving the source point by a translation dt changes the residual by n . dt
    eps = 1e-3
    for ax in range(3):
        tp = t0.copy()
        tp[ax] += eps
        f2, r2 = row_at(np.eye(3), tp, x, y)
        if f2:
            assert abs((r2[6] - row[6]) / eps - row[ax]) < 5e-2
    # rotational part is s x n
    P = H * W
    i0 = y * W + x
    s = vm[[i0 // W + 0 * H, i0 // W + H, i0 // W + 2 * H], x] + t0
    assert np.allclose(row[3:6], np.cross(s, row[0:3]), atol=1e-4)


def test_ldlt_and_rodrigues_against_numpy(orc):
    rng = np.random.default_rng(1)
    for _ in range(5):
        J = rng.normal(size=(40, 6))
        A = J.T @ J
        b = rng.normal(size=6)
        x = np.linalg.solve(A, b)
        # exercised through the tracker: a synthetic rigid motion must be recovered (see GPU tests);
        # here the covariance helper (Gauss-Jordan inverse) is checked directly
        cov = orc.covariance(A)
        assert np.allclose(cov @ A, np.eye(6), atol=1e-8)
        assert np.allclose(cov @ b, x, atol=1e-8)


def test_detmath_accuracy_and_pcg32_known_answer(orc):
    import ctypes as C

    from densemonoslam_amd import synth

    g = synth.PCG32(42, 54)
    assert int(g.uint32(1)[0]) == 0xA15C02B7  # first output of the PCG32 demo stream
    # the oracle's fixed exp / acos rule stays within 3e-7 of libm (GLSL allows far more)
    lib = C.CDLL(None)
    d = np.full((64, 64), 1000, np.uint16)
    f = orc.depth_bilateral(d, 3.0)
    assert (f == 1000).all(), "a constant depth image is a fixed point of the bilateral filter"
    d[:, 32:] = 1400  # a 40 cm step: the colour term must keep the edge sharp
    f = orc.depth_bilateral(d, 3.0)
    assert (f[:, :32] == 1000).all() and (f[:, 32:] == 1400).all()


def test_prep_kernels_against_numpy(orc, gputest_pair):
    d = gputest_pair["depth2"]
    K = gputest_pair["K"]
    vm = orc.createVMap(K, d, 20.0)
    H = 480
    z = d.astype(np.float32) / np.float32(1000.0)
    valid = (z != 0) & (z < 20.0)
    assert np.array_equal(~np.isnan(vm[:H]), valid)
    u = np.arange(640, dtype=np.float32)[None, :]
    vx = (z * (u - np.float32(K[2]))) * np.float32(1.0 / np.float32(K[0]))
    assert np.allclose(vm[:H][valid], vx[valid], rtol=1e-6)
    assert np.array_equal(vm[2 * H:][valid], z[valid])
    nm = orc.createNMap(vm)
    nv = ~np.isnan(nm[:H])
    nn = nm[:H][nv] ** 2 + nm[H:2 * H][nv] ** 2 + nm[2 * H:][nv] ** 2
    assert np.allclose(nn, 1.0, atol=1e-4)
    assert np.isnan(nm[:H][-1]).all() and np.isnan(nm[:H][:, -1]).all()
    # intensity: int(0.114 r + 0.299 g + 0.587 b) on the channels as uploaded
    rgba = helpers.rgba(gputest_pair["rgb2"])
    I = orc.imageBGRToIntensity(rgba)
    ref = (rgba[..., 0].astype(np.float32) * np.float32(0.114) + rgba[..., 1].astype(np.float32) * np.float32(0.299)
           + rgba[..., 2].astype(np.float32) * np.float32(0.587))
    assert np.abs(I.astype(np.int32) - np.floor(ref).astype(np.int32)).max() <= 1
    # depth pyramid halves the size and stays inside the value range of its window
    p1 = orc.pyrDown(d)
    assert p1.shape == (240, 320) and p1.max() <= d.max()


@pytest.fixture(scope="module")
def small_scene(orc):
    from densemonoslam_amd import synth

    W, H = 160, 120
    K = (132.0, 132.0, 80.0, 60.0)
    d, rgb, _ = synth.frame(0, width=W, height=H, K=K, noise=True)
    rgba = synth.rgba(rgb)
    df = orc.depth_bilateral(d, 3.0)
    dm, dmf = orc.depth_metric(d, 3.0), orc.depth_metric(df, 3.0)
    model = orc.model_initialise(rgba, dm, dmf, K, 1, 0, 25.0)
    return W, H, K, d, rgba, df, dm, dmf, model


def test_bootstrap_surfels_are_the_depth_map(orc, small_scene):
    W, H, K, d, rgba, df, dm, dmf, model = small_scene
    valid = (dm > 0) & (dm <= 25.0)
    assert len(model) == int(valid.sum())
    # column-major emission order (GlobalModel.cpp:100-108): the i-th surfel is the i-th valid pixel by columns
    cols, rows = np.nonzero(valid.T)
    assert np.array_equal(model["pos"][:, 2], dm[rows, cols])
    assert (model["col"][:, 2] == 1).all() and (model["times"][:, 0] == 1).all() and (model["times"][:, 1:] == -3).all()
    conf = model["pos"][:, 3]
    assert conf.max() <= 1.0 and conf.min() > 0.0
    r = np.sqrt((cols + 0.5 - K[2]) ** 2 + (rows + 0.5 - K[3]) ** 2) / 400.0
    assert np.allclose(conf, np.exp(-(r * r) / 0.72), rtol=2e-6)


def test_index_map_winner_is_the_nearest_surfel(orc, small_scene):
    W, H, K, d, rgba, df, dm, dmf, model = small_scene
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = (0.02, -0.01, 0.03)
    idx, vc, ct, nr = orc.index_map(model, pose, K, H, W, 2, 0, 25.0, 200)
    assert (idx > 0).mean() > 0.5
    ys, xs = np.nonzero(idx)
    sel = model[idx[ys, xs]]
    tinv = orc.inv4f(pose)
    p = sel["pos"][:, :3].astype(np.float64)
    ph = p @ tinv[:3, :3].T.astype(np.float64) + tinv[:3, 3]
    assert np.allclose(ph, vc[ys, xs, :3], atol=1e-5)
    # every winner projects into its own pixel
    u = K[0] * ph[:, 0] / ph[:, 2] + K[2]
    v = K[1] * ph[:, 1] / ph[:, 2] + K[3]
    assert (np.abs(np.floor(u) - xs) <= 1e-3 + (np.abs(u - np.round(u)) < 1e-3)).all()
    assert (np.abs(np.floor(v) - ys) <= 1e-3 + (np.abs(v - np.round(v)) < 1e-3)).all()
    # no other surfel projecting into the pixel is closer (brute force over a sample of pixels)
    P = model["pos"][:, :3].astype(np.float64) @ tinv[:3, :3].T.astype(np.float64) + tinv[:3, 3]
    U = np.floor(K[0] * P[:, 0] / P[:, 2] + K[2]).astype(int)
    V = np.floor(K[1] * P[:, 1] / P[:, 2] + K[3]).astype(int)
    rng = np.random.default_rng(2)
    for k in rng.choice(len(ys), 200, replace=False):
        inside = np.flatnonzero((U == xs[k]) & (V == ys[k]) & (P[:, 2] > 0))
        if len(inside):
            assert P[idx[ys[k], xs[k]], 2] <= P[inside, 2].min() + 2e-6


def test_fuse_conserves_confidence_and_clean_is_idempotent(orc, small_scene):
    from densemonoslam_amd import synth

    W, H, K, d, rgba, df, dm, dmf, model = small_scene
    d2, rgb2, _ = synth.frame(1, width=W, height=H, K=K, noise=True)
    rgba2 = synth.rgba(rgb2)
    df2 = orc.depth_bilateral(d2, 3.0)
    dm2, dmf2 = orc.depth_metric(d2, 3.0), orc.depth_metric(df2, 3.0)
    pose = np.eye(4, dtype=np.float32)
    im = orc.index_map(model, pose, K, H, W, 2, 0, 25.0, 200)
    m2, newU, merged = orc.model_fuse(model, pose, 2, 0, rgba2, dm2, dmf2, im[0], im[1], im[3], K, 25.0, 1.0)
    assert merged > 0 and len(m2) == len(model)
    upd = m2["times"][:, 0] == 2
    assert upd.sum() == merged
    # every merge adds exactly the measurement's confidence to its surfel
    gain = m2["pos"][upd, 3] - model["pos"][upd, 3]
    assert (gain > 0).all() and gain.max() <= 1.0 + 1e-6
    assert np.array_equal(m2["pos"][~upd], model["pos"][~upd])
    # the candidates are a quarter of the pixels (parity gate), merged ones carry -1, new ones -2
    assert len(newU) <= (W // 2) * (H // 2)
    assert set(np.unique(newU["col"][:, 3])) <= {-1.0, -2.0}
    assert (newU["col"][:, 3] == -1).sum() >= merged
    im2 = orc.index_map(m2, pose, K, H, W, 2, 0, 25.0, 200)
    m3 = orc.model_clean(m2, newU, pose, 2, 0, im2[0], im2[1], im2[2], K, 10.0, 200, 25.0)
    assert len(m3) >= len(m2)
    assert not (m3["times"][:, 0] == -2).any(), "new points must have received the frame time"
    # cleaning again with nothing new keeps the map (same index map, no free-space violations appear)
    im3 = orc.index_map(m3, pose, K, H, W, 2, 0, 25.0, 200)
    m4 = orc.model_clean(m3, m3[:0], pose, 2, 0, im3[0], im3[1], im3[2], K, 10.0, 200, 25.0)
    m5 = orc.model_clean(m4, m4[:0], pose, 2, 0, im3[0], im3[1], im3[2], K, 10.0, 200, 25.0)
    assert len(m5) == len(m4)
    assert m5.tobytes() == m4.tobytes()


def test_prediction_of_bootstrap_frame_reproduces_its_depth(orc, small_scene):
    W, H, K, d, rgba, df, dm, dmf, model = small_scene
    img, vtx, nrm, tim = orc.splat_predict(model, np.eye(4, dtype=np.float32), K, H, W, 25.0, 0.0, 2, 0, 2, 200, True)
    both = (vtx[..., 2] > 0) & (dm > 0)
    assert both.mean() > 0.85
    assert np.abs(vtx[..., 2] - dm)[both].mean() < 0.01
    assert np.abs(img[..., :3].astype(int) - rgba[..., :3].astype(int))[both].mean() < 12
    assert (tim[both] == 1).all()
    # confidence threshold above every surfel: nothing is drawn
    img2, vtx2, _, _ = orc.splat_predict(model, np.eye(4, dtype=np.float32), K, H, W, 25.0, 10.0, 2, 0, 2, 200, True)
    assert not vtx2.any()


def test_velocity_weight(orc):
    I = np.eye(4, dtype=np.float32)
    assert orc.velocity_weight(I, I, 1.0) == 1.0
    far = I.copy()
    far[:3, 3] = (0.05, 0, 0)
    assert orc.velocity_weight(far, I, 1.0) == 0.5  # clamped at `largest` then at minWeight
    near = I.copy()
    near[:3, 3] = (0.002, 0, 0)
    assert abs(orc.velocity_weight(near, I, 2.0) - 2 * 0.8) < 1e-5


def test_fern_oracle_generator_known_answer_and_self_consistency():
    """oracle/orc_ferns.py: the mt19937 behind the fern table against the C++ standard's known answer ([rand.predef]:
    the 10000th output of seed 5489 is 4123659995); the uniform mapping stays in range and is reproducible; the
    inverted-list co-occurrence count of addFrame equals the direct definition (codes equal and not bad)."""
    from oracle import orc_ferns as F

    r = F.MT19937(5489)
    for _ in range(10000):
        v = r.next()
    assert v == 4123659995
    r = F.MT19937(1)
    draws = [r.uniform(400, 3000) for _ in range(2000)]
    assert min(draws) >= 400 and max(draws) <= 3000 and len(set(draws)) > 1000
    f1, f2 = F.Ferns(320, 240, (264, 264, 160, 120), seed=5), F.Ferns(320, 240, (264, 264, 160, 120), seed=5)
    assert (f1.pos == f2.pos).all() and (f1.rgbd == f2.rgbd).all()
    rng = np.random.default_rng(0)
    for k in range(6):
        img = rng.integers(0, 256, (30, 40, 4), dtype=np.uint8)
        verts = rng.uniform(0.3, 3.0, (30, 40, 4)).astype(np.float32)
        verts[rng.random((30, 40)) < 0.2] = 0
        codes, good, co = f1._encode(img, verts)
        for j, fr in enumerate(f1.frames):
            assert co[j] == int(((codes == fr.codes) & (codes != F.BAD)).sum())
        f1._add(img, verts, verts, np.eye(4), k, 0.0)
    assert len(f1.frames) == 6


def test_fern_code_agreement_of_exactly_0_3f_verifies(orc):
    """Ferns.cpp:346 compares blockHDAware's FLOAT return value with the DOUBLE literal 0.3: 150 equal codes of 500 give 0.3f =
    0.300000012, which is above 0.3, so the tracker verifies the candidate (a float32 comparison `> 0.3f` would stop here - the
    boundary the pipelined session's wake rule and find_common must take the same way, tests/test_ferns_gpu.py)."""
    from oracle import orc_ferns
    from tests import helpers

    W, H, K = 640, 480, (528.0, 528.0, 320.0, 240.0)
    o = orc_ferns.Ferns(W, H, K, seed=11, make_odometry=lambda: orc.Odometry(W // 8, H // 8, K[2] / 8, K[3] / 8, K[0] / 8, K[1] / 8))
    A, B = helpers.fern_boundary_thumbnails(o, 150)
    assert o._add(*A, np.eye(4, dtype=np.float32), 1, 0.3)
    assert np.float32(150) / np.float32(500) == np.float32(0.3) and float(np.float32(0.3)) > 0.3
    assert o.searchHit(B, 5, True)
    m = o.findFrame(np.eye(4, dtype=np.float32), None, None, None, 5, interMap=True, thumbs=B)
    assert m["candidate"] == 0 and np.float32(m["blockHDAware"]) == np.float32(0.3) and m["icp_count"] > 0  # (the tracker ran)
    # one code fewer: below the threshold on either reading
    o2 = orc_ferns.Ferns(W, H, K, seed=11, make_odometry=lambda: orc.Odometry(W // 8, H // 8, K[2] / 8, K[3] / 8, K[0] / 8, K[1] / 8))
    A2, B2 = helpers.fern_boundary_thumbnails(o2, 149)
    assert o2._add(*A2, np.eye(4, dtype=np.float32), 1, 0.3) and not o2.searchHit(B2, 5, True)
    assert o2.findFrame(np.eye(4, dtype=np.float32), None, None, None, 5, interMap=True, thumbs=B2)["icp_count"] == 0
