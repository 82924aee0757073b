"""The tracker's scalar section (6x6 solve, exp map, pose update, projection parameters; SO3 update) is one fixed sequence
of IEEE operations (csrc/gn_scalar.hpp).  Here the product's own source, compiled for the host inside libdmslam_hip.so
(dms_debug_scalar_*), is run against the oracle's C restatement of the same sequence (oracle/orc_scalar.c): identical bits
on random well-conditioned, ill-conditioned and singular systems — no GPU involved.  The GPU runs the same source."""
import ctypes as C
import zlib

import numpy as np
import pytest


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rot(rng, ang):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _sums(rng, scale, n=400, rank=7):
    """29 float sums of n random rows whose Jacobian columns have the given scales (rank < 6: a degenerate system)"""
    rows = rng.normal(size=(n, 7)) * scale
    if rank < 6:
        rows[:, rank:6] = 0
    s = []
    for i in range(6):
        for j in range(i, 7):
            s.append(np.sum(rows[:, i] * rows[:, j]))
    s.append(np.sum(rows[:, 6] ** 2))
    s.append(float(n))
    return np.array(s, np.float32)


def _product_gn(lib, si, sr, w, Rprev, tprev, Rt, cam, lvl):
    Rt = Rt.copy()
    A, b = np.zeros(36), np.zeros(6)
    Rc, tc, krk, kt = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
    rc = lib.dms_debug_scalar_gn(_p(si) if si is not None else None, _p(sr) if sr is not None else None, w, _p(Rprev), _p(tprev), _p(Rt),
                                 cam[0], cam[1], cam[2], cam[3], lvl, _p(A), _p(b), _p(Rc), _p(tc), _p(krk), _p(kt))
    assert rc == 0
    return Rt, A, b, Rc, tc, krk, kt


CASES = [
    ("well conditioned", np.array([1, 1, 1, 2, 2, 2, 0.05]), np.array([30, 30, 30, 40, 40, 40, 0.3]), 7),
    ("rotation weak", np.array([1, 1, 1, 1e-3, 1e-3, 1e-3, 0.05]), np.array([30, 30, 30, 1e-2, 1e-2, 1e-2, 0.3]), 7),
    ("rank 4 (pivoted path)", np.array([1, 1, 1, 2, 2, 2, 0.05]), np.array([30, 30, 30, 40, 40, 40, 0.3]), 4),
]


@pytest.mark.parametrize("name,sc_icp,sc_rgb,rank", CASES)
@pytest.mark.parametrize("terms", ["both", "icp", "rgb"])
def test_gn_update_product_host_build_equals_oracle_restatement(orc, name, sc_icp, sc_rgb, rank, terms):
    from densemonoslam_amd.capi import lib

    rng = np.random.default_rng(zlib.crc32((name + terms).encode()))  # not hash(): str hashes differ per process
    cam = (528.0, 528.0, 320.0, 240.0)
    for trial in range(40):
        si = _sums(rng, sc_icp, rank=rank) if terms in ("both", "icp") else None
        sr = _sums(rng, sc_rgb, rank=rank) if terms in ("both", "rgb") else None
        Rprev = _rot(rng, rng.uniform(0, 1.0)).astype(np.float32).reshape(9)
        tprev = rng.normal(size=3).astype(np.float32)
        Rt = np.eye(4)
        Rt[:3, :3] = _rot(rng, rng.uniform(0, 0.05))
        Rt[:3, 3] = rng.normal(size=3) * 0.02
        Rt = Rt.reshape(16)
        lvl = trial % 3
        got = _product_gn(lib, si, sr, 10.0, Rprev, tprev, Rt, cam, lvl)
        want = orc.scalar_gn_update(si, sr, 10.0, Rprev, tprev, Rt, cam, lvl)
        for g, w_, what in zip(got, want, ("resultRt", "A", "b", "Rcurr", "tcurr", "krkinv", "kt")):
            assert np.asarray(g).tobytes() == np.asarray(w_).tobytes(), (name, terms, trial, what, g, w_)
        if rank == 7:  # a real update happened
            assert np.isfinite(got[3]).all() and not np.array_equal(got[0], Rt)


def test_gn_update_all_zero_system_is_the_identity_update(orc):
    from densemonoslam_amd.capi import lib

    cam = (528.0, 528.0, 320.0, 240.0)
    z = np.zeros(29, np.float32)
    Rprev = np.eye(3, dtype=np.float32).reshape(9)
    tprev = np.array([0.1, 0.2, 0.3], np.float32)
    Rt = np.eye(4).reshape(16)
    got = _product_gn(lib, z, z, 10.0, Rprev, tprev, Rt, cam, 0)
    want = orc.scalar_gn_update(z, z, 10.0, Rprev, tprev, Rt, cam, 0)
    for g, w_ in zip(got, want):
        assert np.asarray(g).tobytes() == np.asarray(w_).tobytes()
    assert np.array_equal(got[0], Rt) and np.array_equal(got[4], tprev)


def test_so3_update_product_host_build_equals_oracle_restatement(orc):
    from densemonoslam_amd.capi import lib

    rng = np.random.default_rng(7)
    cam = (528.0, 528.0, 320.0, 240.0)
    for trial in range(60):
        rows = rng.normal(size=(300, 4)) * np.array([2000, 2000, 300, 20])
        if trial % 10 == 9:
            rows[:, 2] = 0  # degenerate column: pivoted solve with a zero pivot
        s = []
        for i in range(3):
            for j in range(i, 4):
                s.append(np.sum(rows[:, i] * rows[:, j]))
        s += [np.sum(rows[:, 3] ** 2), 300.0]
        s = np.array(s, np.float32)
        R_lr = _rot(rng, rng.uniform(0, 0.05)).astype(np.float32).reshape(9)
        resultR = R_lr.astype(np.float64)
        lr, rr = R_lr.copy(), resultR.copy()
        ib, ki, kr = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32)
        assert lib.dms_debug_scalar_so3(_p(s), _p(lr), _p(rr), cam[0], cam[1], cam[2], cam[3], _p(ib), _p(ki), _p(kr)) == 0
        want = orc.scalar_so3_update(s, R_lr, resultR, cam)
        for g, w_, what in zip((lr, rr, ib, ki, kr), want, ("R_lr", "resultR", "imageBasis", "kinv", "krlr")):
            assert g.tobytes() == w_.tobytes(), (trial, what, g, w_)


def test_canonical_scalar_section_agrees_with_the_independent_restatement(orc, gputest_pair):
    """The oracle's two scalar sections — the canonical operation order (orc_scalar.c) and the independent Eigen-like
    restatement (orc_odometry.c, solve mode 0) — evaluate the same formulas: after ONE Gauss-Newton iteration from the same
    sums (ICP only, one level: nothing has been amplified by a correspondence search yet) the poses agree to float
    rounding."""
    from tests import helpers

    K = gputest_pair["K"]
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    out = []
    for mode in (1, 0):
        o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
        o.setSolveMode(mode)
        o.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        o.initICP(gputest_pair["depth2"], 20.0)
        t, R, res = o.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), False, 100.0, False, True, False)
        out.append((np.array(res.trace[0]), np.array(res.lastA), res.canon_retries))
    first = [np.array(x[0], np.float64) for x in out]
    assert np.abs(first[0] - first[1]).max() < 3e-7, (first[0], first[1])
    assert out[0][2] == out[1][2]


def test_canonical_sums_do_not_depend_on_the_thread_count_or_the_pixel_order(orc):
    rng = np.random.default_rng(3)
    n = 50000
    rows = (rng.normal(size=(n, 7)) * np.array([1, 1, 1, 2, 2, 2, 0.05])).astype(np.float32)
    found = rng.random(n) < 0.8
    rows[~found] = 0
    E0 = np.array([17, 17, 17, 19, 19, 19, 8], np.int32)
    orc.set_threads(1)
    s1, E1, r1 = orc.canon_reduce(rows, found, E0)
    orc.set_threads(7)
    s7, E7, r7 = orc.canon_reduce(rows, found, E0)
    perm = rng.permutation(n)
    sp, Ep, rp = orc.canon_reduce(rows[perm], found[perm], E0)
    assert s1.tobytes() == s7.tobytes() == sp.tobytes() and r1 == r7 == rp == 0
    # against the fp64 sums of the same exact products: below 2^-30 of the Cauchy-Schwarz bound of each value
    k = 0
    d = (rows.astype(np.float64) ** 2).sum(0)
    for i in range(6):
        for j in range(i, 7):
            exact = float((rows[:, i].astype(np.float64) * rows[:, j].astype(np.float64)).sum())
            assert abs(float(s1[k]) - exact) <= 1e-6 * np.sqrt(d[i] * d[j]) + 6e-8 * abs(exact), (i, j)
            k += 1
    assert s1[28] == found.sum()
    # exponents that are too small: the reduction repeats itself on a coarser grid and still agrees to float rounding
    sl, El, rl = orc.canon_reduce(rows, found, E0 - 20)
    assert rl == 3 and (El == E0 - 20 + 24).all()
    assert np.allclose(sl, s1, rtol=1e-6, atol=1e-3)


def test_rodrigues_every_angle_is_canonical_and_accurate(orc):
    """The exp map of the scalar section never calls the math library (sin / cos differ between the device's, clang's host
    and glibc's): polynomial coefficients below 0.77 rad, Cody-Waite reduction + the same kernels above.  The oracle's
    restatement against scipy's rotation vectors at every magnitude a degenerate system can produce, and the product's
    host build against the oracle bit for bit on systems whose update is a large rotation."""
    from scipy.spatial.transform import Rotation

    from densemonoslam_amd.capi import lib
    from oracle import orc as o

    rng = np.random.default_rng(11)
    for mag in (1e-17, 1e-9, 1e-3, 0.3, 0.76, 0.78, 1.5, 3.1, 3.2, 6.3, 40.0, 1234.5, 9e4):
        for _ in range(20):
            r = rng.normal(size=3)
            r *= mag / np.linalg.norm(r)
            R = np.zeros(9)
            o.lib.orc_scalar_rodrigues(_p(r), _p(R))
            want = Rotation.from_rotvec(r).as_matrix().reshape(9) if mag >= 2.3e-16 else np.eye(3).reshape(9)
            assert np.abs(R - want).max() < 4e-16 * max(1.0, mag), (mag, R, want)
    bad_angle = 0
    cam = (528.0, 528.0, 320.0, 240.0)
    for trial in range(300):
        si = _sums(rng, np.array([1, 1, 1, 1e-3, 1e-3, 1e-3, 0.05]))
        Rprev = _rot(rng, rng.uniform(0, 1.0)).astype(np.float32).reshape(9)
        tprev = rng.normal(size=3).astype(np.float32)
        Rt = np.eye(4).reshape(16)
        got = _product_gn(lib, si, None, 10.0, Rprev, tprev, Rt, cam, trial % 3)
        want = orc.scalar_gn_update(si, None, 10.0, Rprev, tprev, Rt, cam, trial % 3)
        for g, w_ in zip(got, want):
            assert np.asarray(g).tobytes() == np.asarray(w_).tobytes()
        ang = np.arccos(np.clip((np.trace(got[0].reshape(4, 4)[:3, :3]) - 1) / 2, -1, 1))
        bad_angle += ang > 0.77
    assert bad_angle > 30, "the case is meant to reach the large-angle branch"
