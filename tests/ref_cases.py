"""Cases that pin the oracle's tracking steps to the REFERENCE's own kernels (Cuda/reduce.cu, built by
oracle/ref_build.sh into oracle/_ref/libref_reduce.so and run on an MI355X).

One runner, two back ends with the same four calls (icpStep, computeRgbResidual, rgbStep, so3Step):
  * oracle.ref  -> tests/golden/make_ref_reduce_golden.py records its outputs in tests/golden/ref_reduce.npz (GPU box);
  * oracle.orc  -> tests/test_ref_pin_cpu.py compares the restatement with that file (CPU, every round).
Inputs: the reference's GPUTest RGB-D pair (committed fixture) taken through the tracker's pyramid set-up; the fixture
stores a SHA-256 of every input array, so the CPU test knows it feeds the restatement exactly what the reference saw.

What each output pins:
  * `*_row*`  one pixel / one correspondence alone in the image: the sums ARE that pixel's products (adding zeros is exact),
              so these pin the per-pixel arithmetic bit for bit, free of any summation order;
  * `rgbres_*` every DataTerm field of every pixel plus the two integer sums: exact;
  * `icp_* / rgb_* / so3_*` whole-image sums: inlier counts exact, float sums to summation-order tolerance (the reference
              adds fp32 partials in launch order: thread-strided, wave shuffles, one block of partials).
"""
import hashlib

import numpy as np

from tests import helpers

K = (528.0, 528.0, 320.0, 240.0)
SOBEL_SCALE = 1.0 / 8.0
MIN_GRAD = (5.0, 3.0, 1.0)  # RGBDOdometry.cpp:60-62
DIST_THRES, ANGLE_THRES = 0.10, float(np.sin(np.radians(20.0)))  # RGBDOdometry.cpp:33-34
MAX_DEPTH_DELTA = 0.07  # RGBDOdometry.cpp:38
N_ICP_ROWS, N_RGB_ROWS, N_SO3_ROWS = 192, 128, 128


def _rot(axis, ang):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


# current-frame pose estimates for icpStep (previous pose = identity) and relative motions for the photometric steps
POSES = [(np.eye(3), np.zeros(3)),
         (_rot((0.2, 1.0, 0.1), np.radians(0.5)), np.array([0.004, -0.002, 0.003])),
         (_rot((1.0, -0.3, 0.5), np.radians(2.0)), np.array([-0.015, 0.010, 0.012]))]


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.view(np.uint8).reshape(-1).tobytes()).hexdigest()


def inputs(orc, pair):
    """Per level: the arrays RGBDOdometry.cpp:443-539 hands to the four steps, from the oracle's pyramid set-up of the
    GPUTest pair (the set-up itself is cudafuncs.cu's, not pinned here; these are just realistic inputs)."""
    verts, norms = helpers.gputest_model_maps(pair["depth1_raw"], K)
    o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
    o.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
    o.initRGBModel(helpers.rgba(pair["rgb1"]))
    o.initICP(pair["depth2"], 20.0)
    o.initRGB(helpers.rgba(pair["rgb2"]))
    o.initFirstRGB(helpers.rgba(pair["rgb1"]))
    lv = []
    for lvl in range(3):
        d = dict(vmap_curr=o.buffer(0, lvl), nmap_curr=o.buffer(1, lvl), vmap_g_prev=o.buffer(2, lvl), nmap_g_prev=o.buffer(3, lvl),
                 lastDepth=o.buffer(4, lvl), nextDepth=o.buffer(5, lvl), lastImage=o.buffer(6, lvl), nextImage=o.buffer(7, lvl),
                 lastNextImage=o.buffer(8, lvl))
        d["dIdx"], d["dIdy"] = orc.computeDerivativeImages(d["nextImage"])
        d["cloud"] = orc.projectToPointCloud(d["lastDepth"], K, lvl)
        lv.append(d)
    return lv


def input_hashes(lv):
    return {"in_L%d_%s" % (l, k): np.array(sha(v)) for l, d in enumerate(lv) for k, v in d.items()}


def _cam(lvl):
    return np.array([np.float32(v) / np.float32(1 << lvl) for v in K], np.float32)


def _photo_args(lvl, R, t):
    cam = _cam(lvl).astype(np.float64)
    Km = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float64)
    krkinv = (Km @ R @ np.linalg.inv(Km)).astype(np.float32)
    kt = (Km @ t).astype(np.float32)
    return Km, kt, krkinv


def _se3_vec(A, b, res=None):
    iu = np.triu_indices(6)
    v = [np.asarray(A)[iu], np.asarray(b)]
    if res is not None:
        v.append(np.asarray(res))
    return np.concatenate(v).astype(np.float32)


def run(be, lv, rows_from=None):
    """All cases through back end `be`.  `rows_from`: a finished result dict whose pixel / correspondence choices for the
    single-row cases are re-used (so both back ends isolate the same pixels)."""
    out = {}
    for lvl, d in enumerate(lv):
        cam = _cam(lvl)
        for pi, (R, t) in enumerate(POSES):
            A, b, res = be.icpStep(R.astype(np.float32), t.astype(np.float32), d["vmap_curr"], d["nmap_curr"], np.eye(3, dtype=np.float32),
                                   np.zeros(3, np.float32), cam, d["vmap_g_prev"], d["nmap_g_prev"], DIST_THRES, ANGLE_THRES)
            out["icp_L%d_P%d" % (lvl, pi)] = _se3_vec(A, b, res)
            # photometric: `next` seen from `last`, relative motion = the same small transforms
            Km, kt, krkinv = _photo_args(lvl, R, t)
            minScale = float(MIN_GRAD[lvl] ** 2 / SOBEL_SCALE ** 2)
            co, sigma, count = be.computeRgbResidual(minScale, d["dIdx"], d["dIdy"], d["lastDepth"], d["nextDepth"], d["lastImage"],
                                                     d["nextImage"], MAX_DEPTH_DELTA, kt, krkinv)
            key = "rgbres_L%d_P%d" % (lvl, pi)
            out[key + "_sums"] = np.array([sigma, count], np.int64)
            out[key + "_valid"] = np.packbits(co["valid"] != 0)
            out[key + "_sha"] = np.array(sha(co))
            if lvl == 2:
                out[key + "_corres"] = co
            for si, sg in enumerate((float(np.sqrt(max(count, 1))), -1.0)):
                A, b = be.rgbStep(co, sg, d["cloud"], float(cam[0]), float(cam[1]), d["dIdx"], d["dIdy"], SOBEL_SCALE)
                out["rgb_L%d_P%d_S%d" % (lvl, pi, si)] = _se3_vec(A, b)
            if lvl == 2:
                ib = krkinv
                ki = np.linalg.inv(Km).astype(np.float32)
                kr = (Km @ R).astype(np.float32)
                A, b, res = be.so3Step(d["lastNextImage"], d["nextImage"], ib, ki, kr)
                out["so3_L2_P%d" % pi] = np.concatenate([A[np.triu_indices(3)], b, res]).astype(np.float32)
    # ---- single rows at level 2 (160x120): one pixel / one correspondence alone ----
    d, cam = lv[2], _cam(2)
    rows, cols = d["nextImage"].shape
    R, t = POSES[1]
    if rows_from is None:
        rng = np.random.default_rng(20260930)
        ok = np.argwhere(~np.isnan(d["vmap_curr"][:rows]))
        pix = ok[rng.choice(len(ok), N_ICP_ROWS, replace=False)]
    else:
        pix = rows_from["icp_row_pix"]
    out["icp_row_pix"] = np.asarray(pix, np.int32)
    icp_rows = []
    for (y, x) in pix:
        vm = d["vmap_curr"].copy()
        keep = vm[y::rows, x].copy()
        vm[:rows] = np.nan  # the x plane decides validity (reduce.cu:266)
        vm[y::rows, x] = keep
        A, b, res = be.icpStep(R.astype(np.float32), t.astype(np.float32), vm, d["nmap_curr"], np.eye(3, dtype=np.float32),
                               np.zeros(3, np.float32), cam, d["vmap_g_prev"], d["nmap_g_prev"], DIST_THRES, ANGLE_THRES)
        icp_rows.append(_se3_vec(A, b, res))
    out["icp_rows"] = np.stack(icp_rows)
    co = (rows_from or out)["rgbres_L2_P1_corres"]
    if rows_from is None:
        rng = np.random.default_rng(7)
        okc = np.argwhere(co["valid"] != 0)
        cpix = okc[rng.choice(len(okc), min(N_RGB_ROWS, len(okc)), replace=False)]
    else:
        cpix = rows_from["rgb_row_pix"]
    out["rgb_row_pix"] = np.asarray(cpix, np.int32)
    for si, sg in enumerate((37.5, -1.0)):
        rr = []
        for (y, x) in cpix:
            one = np.zeros_like(co)
            one[y, x] = co[y, x]
            A, b = be.rgbStep(one, sg, d["cloud"], float(cam[0]), float(cam[1]), d["dIdx"], d["dIdy"], SOBEL_SCALE)
            rr.append(_se3_vec(A, b))
        out["rgb_rows_S%d" % si] = np.stack(rr)
    # ---- single SO3 rows: a 3 x 3 image has ONE pixel that passes so3Step's bounds test (reduce.cu:975-979), so the sums of
    # a call on a 3 x 3 patch pair are that pixel's ten products and its flag.  Patches cut from the level-2 images, the
    # level-2 camera matrices with a small rotation (the warped pixel stays (1, 1)).
    rng = np.random.default_rng(11)
    rows, cols = lv[2]["nextImage"].shape
    cam = _cam(2).astype(np.float64)
    Km = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float64)
    so3_rows = []
    for i in range(N_SO3_ROWS):
        y, x = int(rng.integers(1, rows - 2)), int(rng.integers(1, cols - 2))
        last = np.ascontiguousarray(lv[2]["lastNextImage"][y - 1:y + 2, x - 1:x + 2])
        nxt = np.ascontiguousarray(lv[2]["nextImage"][y - 1:y + 2, x - 1:x + 2])
        Rs = _rot(rng.standard_normal(3), np.radians(rng.uniform(0.0, 0.05)))
        ib = (Km @ Rs @ np.linalg.inv(Km)).astype(np.float32)
        ib[:, 2] = [np.float32(1.0) - ib[0, 0] - ib[0, 1], np.float32(1.0) - ib[1, 0] - ib[1, 1], np.float32(1.0) - ib[2, 0] - ib[2, 1]]  # (1, 1, 1) -> ~(1, 1, 1)
        A, b, res = be.so3Step(last, nxt, ib, np.linalg.inv(Km).astype(np.float32), (Km @ Rs).astype(np.float32))
        so3_rows.append(np.concatenate([A[np.triu_indices(3)], b, res]).astype(np.float32))
    out["so3_rows"] = np.stack(so3_rows)
    return out


# ---- whole tracker calls: the restatement's host loop around the reference's kernels -------------------------------------------
TRACKER_CONFIGS = {
    "C2_icp_fast": dict(rgbOnly=False, icpWeight=100.0, pyramid=False, fastOdom=True, so3=False),
    "C3_full": dict(rgbOnly=False, icpWeight=10.0, pyramid=True, fastOdom=False, so3=True),
    "gputest": dict(rgbOnly=False, icpWeight=10.0, pyramid=False, fastOdom=False, so3=True),  # GPUTest.cpp:278
    "rgb_only": dict(rgbOnly=True, icpWeight=10.0, pyramid=True, fastOdom=False, so3=False),
    "icp_pyramid": dict(rgbOnly=False, icpWeight=100.0, pyramid=True, fastOdom=False, so3=False),
}


def run_trackers(orc, pair, hooks=None, fused=False):
    """getIncrementalTransformation from the identity on the GPUTest pair (the harness protocol, GPUTest.cpp:247-286) for every
    configuration, sums in plain order (sum mode 0), rows without contraction; `hooks`: addresses from oracle.ref.step_hooks()."""
    out = {}
    for name, cfg in TRACKER_CONFIGS.items():
        verts, norms = helpers.gputest_model_maps(pair["depth1_raw"], K)
        o = orc.Odometry(640, 480, K[2], K[3], K[0], K[1])
        o.setSumMode(False)
        o.setFusedRows(fused)
        if hooks:
            o.setStepHooks(*hooks)
        o.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        o.initRGBModel(helpers.rgba(pair["rgb1"]))
        o.initICP(pair["depth2"], 20.0)
        o.initRGB(helpers.rgba(pair["rgb2"]))
        o.initFirstRGB(helpers.rgba(pair["rgb1"]))
        t, R, res = o.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **cfg)
        out["trk_%s_t" % name] = np.asarray(t, np.float32)
        out["trk_%s_R" % name] = np.asarray(R, np.float32)
        out["trk_%s_iters" % name] = np.array([res.so3_iterations_run] + list(res.iterations_run), np.int32)
        out["trk_%s_counts" % name] = np.array([res.lastICPCount, res.lastRGBCount, res.lastSO3Count], np.float64)
        out["trk_%s_errors" % name] = np.array([res.lastICPError, res.lastRGBError, res.lastSO3Error], np.float64)
        out["trk_%s_trace" % name] = np.array([list(res.trace[i]) for i in range(res.trace_len)], np.float32).reshape(-1, 3, 4)
    return out
