"""Shared test helpers (host-side numpy only)."""
import numpy as np


def gputest_model_maps(depth_raw, K):
    """Model vertex / normal RGBA32F images from frame 1, built on the CPU the way the reference
    harness does (GPUTest/src/GPUTest.cpp:69-129): forward differences, TUM depth / 5000, zero
    where any 4-neighbour is missing, border rows/cols left zero."""
    fx, fy, cx, cy = K
    d = depth_raw.astype(np.float32) / np.float32(5000.0)
    H, W = d.shape
    rows, cols = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ifx, ify = np.float32(1.0) / np.float32(fx), np.float32(1.0) / np.float32(fy)
    vx = ((cols - np.float32(cx)) * d) * ifx
    vy = ((rows - np.float32(cy)) * d) * ify
    V = np.stack([vx, vy, d], axis=-1).astype(np.float32)
    verts = np.zeros((H, W, 4), np.float32)
    norms = np.zeros((H, W, 4), np.float32)
    r = depth_raw
    ok = np.zeros((H, W), bool)
    ok[1:-1, 1:-1] = (r[1:-1, 1:-1] > 0) & (r[2:, 1:-1] > 0) & (r[1:-1, 2:] > 0) & (r[:-2, 1:-1] > 0) & (r[1:-1, :-2] > 0)
    del_x = np.zeros_like(V)
    del_y = np.zeros_like(V)
    del_x[:, :-1] = V[:, 1:] - V[:, :-1]
    del_y[:-1, :] = V[1:, :] - V[:-1, :]
    n = np.cross(del_x, del_y).astype(np.float32)
    nn = np.sqrt((n * n).sum(-1, keepdims=True))
    with np.errstate(invalid="ignore", divide="ignore"):
        n = n / nn
    # the harness carries the previous pixel's point/normal over invalid pixels? No: it resets
    # both to (0,0,0,1) (GPUTest.cpp:106-110).
    verts[..., :3] = np.where(ok[..., None], V, 0)
    norms[..., :3] = np.where(ok[..., None], n, 0)
    verts[1:-1, 1:-1, 3] = 1
    norms[1:-1, 1:-1, 3] = 1
    return verts, norms


def rgba(rgb):
    return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)


def rot_angle_deg(Ra, Rb):
    """Angle of Ra·Rbᵀ.  atan2 of (|skew part|, (trace-1)/2): well-conditioned at small angles
    (arccos of the trace alone cannot resolve below ~0.03° on float32 matrices)."""
    R = np.asarray(Ra, np.float64).reshape(3, 3) @ np.asarray(Rb, np.float64).reshape(3, 3).T
    s = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    c = (np.trace(R) - 1.0) / 2.0
    return float(np.degrees(np.arctan2(np.linalg.norm(s), c)))


def assert_pose_close(t_a, R_a, t_b, R_b, tol_m=0.0, tol_deg=0.0, what=""):
    """Poses of the tracker object / the frame step: IDENTICAL by default since round 3 (canonical sums + canonical scalar
    section: tests/test_scalar_cpu.py, csrc/canon.hpp) — far inside the north-star bar of 1 mm / 0.01 degree per step.
    A caller that compares against something other than the oracle's canonical modes passes its tolerance."""
    dt = float(np.linalg.norm(np.asarray(t_a, np.float64) - np.asarray(t_b, np.float64)))
    da = rot_angle_deg(R_a, R_b)
    assert dt <= tol_m, "%s translation differs by %.3e m (> %.1e)" % (what, dt, tol_m)
    assert da <= tol_deg, "%s rotation differs by %.3e deg (> %.1e)" % (what, da, tol_deg)
    return dt, da


def assert_pose_identical(t_a, R_a, t_b, R_b, what=""):
    """The tracker object's bar since round 3: the same bits (NaN for NaN).  Canonical sums + canonical scalar section make the
    pose a pure function of the inputs on both sides; anything else is a real divergence, not summation noise."""
    ta, tb = np.asarray(t_a, np.float32).reshape(-1), np.asarray(t_b, np.float32).reshape(-1)
    Ra, Rb = np.asarray(R_a, np.float32).reshape(-1), np.asarray(R_b, np.float32).reshape(-1)
    if ta.tobytes() != tb.tobytes() or Ra.tobytes() != Rb.tobytes():
        if not (np.isnan(ta).any() or np.isnan(tb).any()):
            dt = float(np.linalg.norm(ta.astype(np.float64) - tb.astype(np.float64)))
            raise AssertionError("%s poses differ: translation by %.3e m, rotation by %.3e deg" % (what, dt, rot_angle_deg(Ra, Rb)))
        assert nan_equal(ta, tb) and nan_equal(Ra, Rb), "%s poses differ (NaN pattern)" % what


def nan_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def planes_equal_where_valid(a, b):
    """Stacked-plane maps: plane x must match including the NaN pattern; planes y/z only where
    plane x is valid (invalid pixels keep stale data there)."""
    a, b = np.asarray(a), np.asarray(b)
    H = a.shape[0] // 3
    ax, bx = a[:H], b[:H]
    if not nan_equal(ax, bx):
        return False
    ok = ~np.isnan(ax)
    return bool((a[H:2 * H][ok] == b[H:2 * H][ok]).all() and (a[2 * H:][ok] == b[2 * H:][ok]).all())


def fern_boundary_thumbnails(o, equal=150):
    """Two W/8 x H/8 thumbnail sets (RGBA8 image, RGBA32F vertex, RGBA32F normal) whose descriptors under the fern table of `o`
    (an oracle orc_ferns.Ferns) are valid in all o.num ferns and EQUAL in exactly `equal` of them: blockHDAware = equal / num.
    With 150 of 500 the float ratio rounds to 0.3f, which Ferns.cpp:346 compares with the DOUBLE literal 0.3 - it passes there."""
    th, tw = o.height, o.width
    v = np.zeros((th, tw, 4), np.float32)
    v[..., 2] = 1.0  # 1000 mm everywhere: every fern valid, the depth bit the same in both frames
    v[..., 3] = 1.0
    n = np.zeros((th, tw, 4), np.float32)
    n[..., 2] = -1.0
    a = np.zeros((th, tw, 4), np.uint8)
    a[..., 3] = 255
    ca, ga, _ = o._encode(a, v)
    assert ga == o.num
    full = a.copy()
    full[..., :3] = 255
    cf, _, _ = o._encode(full, v)
    per_pixel = {}
    for i in range(o.num):  # ferns whose code changes when their pixel turns white, grouped by pixel
        if cf[i] != ca[i]:
            per_pixel.setdefault((int(o.pos[i, 1]), int(o.pos[i, 0])), []).append(i)
    want = o.num - equal
    b = a.copy()
    for (y, x), fs in sorted(per_pixel.items(), key=lambda kv: -len(kv[1])):
        if len(fs) <= want:
            b[y, x, :3] = 255
            want -= len(fs)
    assert want == 0, "could not place the differing ferns"
    cb, gb, _ = o._encode(b, v)
    assert gb == o.num and int((ca == cb).sum()) == equal
    return (a, v, n), (b, v.copy(), n.copy())
