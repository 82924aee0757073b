"""Cases that pin the pyramid / preparation operators and the NID scores to the REFERENCE's own
Cuda/cudafuncs.cu, built by oracle/ref_build.sh into oracle/_ref/libref_cudafuncs.so and run on an MI355X.

One chain, three back ends with the same operator names:
  * oracle.ref_cf           -> tests/golden/make_ref_cudafuncs_golden.py records its outputs in tests/golden/ref_cudafuncs.npz;
  * OrcOps (the restatement, oracle/orc_track.c + oracle/orc_nid.py) -> tests/test_ref_cf_pin_cpu.py, every round on the CPU;
  * HipOps (the product's operator layer through the C ABI)          -> tests/test_ref_cf_pin_gpu.py.
The chain is the sequence of operator calls RGBDOdometry::init* makes (Utils/RGBDOdometry.cpp:118-292, :412), plus the
key-frame calls (KeyFrame.h:46-219: vertex-only copyMaps / tranformMaps, verticesToDepth on a 3-plane map) and the two NID
scores (MutualInformation.cpp:154-213), on
  * "full":  the reference's GPUTest RGB-D pair at 640 x 480 (K = 528, 528, 320, 240);
  * "small": the same pair sub-sampled and cropped to a ragged 67 x 49 (odd in both directions, so every halving truncates).

How an output is recorded: arrays of at most FULL_MAX elements in full; larger ones as a SHA-256 of the canonical bytes,
a SHA-256 of the NaN mask and 4096 sampled values (fixed positions).  Canonical = every NaN is the same NaN, and the y / z
planes of a 3-plane map are zero where the x plane is NaN (the reference leaves them unwritten there).
With a `feed` (the fixture), a stage whose reference output was recorded in full takes THAT as its input instead of the
back end's own previous output, so every operator is compared in isolation on the reference's inputs ("small" case); where
only hashes exist the back end's own outputs cascade ("full" case).

The one operator the reference cannot run here is imageBGRToIntensity (texture sampler); the intensity images below are
inputs, produced by the restatement and recorded by hash.

Tolerances: every operator is compared bit for bit, except where the reference calls rsqrtf (operators.cuh:79-83
`normalized`, used by createNMap and resizeNMap): an approximate instruction on every GPU (v_rsq_f32 on gfx950, 1 ulp;
rsqrt.approx on NVIDIA, 2 ulp), where the restatement and the product divide by a correctly rounded square root.  Those two
(and what is computed from their outputs without a feed) are held to NORMAL_TOL on each component, NaN masks exact.
"""
import hashlib

import numpy as np

from tests import helpers

FULL_MAX = 3 * 49 * 67  # one 3-plane map of the small case
NORMAL_TOL = 4e-7        # 1 ulp of rsqrt on a unit vector's component (2^-23 = 1.2e-7) with margin for the products
N_SAMPLE = 4096
K_FULL = (528.0, 528.0, 320.0, 240.0)
CROP = (slice(25, 25 + 49), slice(30, 30 + 67))  # the small case's window in the 5x sub-sampled pair
CUTOFF, MAX_DEPTH_RGB = 20.0, 6.0  # GPUTest.cpp:215-216 / RGBDOdometry.cpp:37


def _rot(axis, ang):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx


POSE_R = _rot((0.3, 1.0, -0.2), np.radians(7.0)).astype(np.float32)
POSE_T = np.array([0.31, -0.12, 0.45], np.float32)


# ---- canonical form, hashing, comparison ------------------------------------------------------------------------------------
def canon(a, planes3=False):
    a = np.array(a, copy=True)
    if a.dtype.kind == "f":
        if planes3:
            H = a.shape[0] // 3
            bad = np.isnan(a[:H])
            a[H:2 * H][bad] = 0
            a[2 * H:][bad] = 0
        a[np.isnan(a)] = np.float32(np.nan)
    return np.ascontiguousarray(a)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).reshape(-1).tobytes()).hexdigest()


def _sample_idx(n):
    return np.random.default_rng(n).integers(0, n, N_SAMPLE)


class Rec:
    """Collects the chain's outputs; with `feed`, hands the fixture's full arrays downstream."""

    def __init__(self, feed=None):
        self.out, self.feed, self.planes3 = {}, feed, {}

    def __call__(self, name, arr, planes3=False):
        arr = canon(np.asarray(arr), planes3)
        self.out[name] = arr
        self.planes3[name] = planes3
        if self.feed is not None and name in self.feed:
            return np.asarray(self.feed[name])
        return arr


def pack(out):
    """Result dict -> what the fixture stores."""
    z = {}
    for k, a in out.items():
        a = np.asarray(a)
        if a.size <= FULL_MAX:
            z[k] = a
        else:
            z[k + "__sha"] = np.array(sha(a))
            z[k + "__nansha"] = np.array(sha(np.packbits(np.isnan(a)))) if a.dtype.kind == "f" else np.array("")
            z[k + "__sample"] = a.reshape(-1)[_sample_idx(a.size)]
    return z


def compare(name, got, fx, tol=0.0):
    """`got` (canonical array) against the fixture entry.  Returns "exact" or "tol"; raises with a diagnostic."""
    got = np.asarray(got)
    if name in fx:
        want = np.asarray(fx[name])
        assert want.shape == got.shape and want.dtype == got.dtype, (name, want.shape, got.shape, want.dtype, got.dtype)
        if want.tobytes() == got.tobytes():
            return "exact"
        if got.dtype.kind == "f":
            assert (np.isnan(want) == np.isnan(got)).all(), "%s: NaN masks differ at %d places" % (name, int((np.isnan(want) != np.isnan(got)).sum()))
            d = np.nan_to_num(np.abs(want.astype(np.float64) - got.astype(np.float64)))
        else:
            d = np.abs(want.astype(np.int64) - got.astype(np.int64))
        assert d.max() <= tol, "%s: %d of %d elements differ, worst |d| = %.3e (tolerance %.1e)" % (name, int((d > 0).sum()), d.size, d.max(), tol)
        return "tol"
    assert name + "__sha" in fx, "fixture has no entry " + name
    if str(fx[name + "__sha"]) == sha(got):
        return "exact"
    want = np.asarray(fx[name + "__sample"])
    mine = got.reshape(-1)[_sample_idx(got.size)]
    if got.dtype.kind == "f":
        assert str(fx[name + "__nansha"]) == sha(np.packbits(np.isnan(got))), name + ": NaN masks differ"
        d = np.nan_to_num(np.abs(want.astype(np.float64) - mine.astype(np.float64)))
    else:
        d = np.abs(want.astype(np.int64) - mine.astype(np.int64))
    assert tol > 0, "%s: hashes differ (exact comparison); of %d sampled elements %d differ, worst |d| = %.3e" % (name, d.size, int((d > 0).sum()), d.max())
    assert d.max() <= tol, "%s: sampled elements differ by up to %.3e (tolerance %.1e)" % (name, d.max(), tol)
    return "tol"


# which outputs carry the rsqrt tolerance: createNMap's and resizeNMap's, and - when a back end's own outputs cascade (no feed) -
# what is computed from them (tranformMaps of the level-1 / level-2 normals: three products of a rotation row, 2 x the tolerance)
def tol_of(name, fed):
    if name.startswith("nmap_L") or name in ("mnmap_L1", "mnmap_L2"):
        return NORMAL_TOL
    if name in ("tnmap_L1", "tnmap_L2") and not fed:
        return 2 * NORMAL_TOL
    return 0.0


# ---- inputs -----------------------------------------------------------------------------------------------------------------
def inputs(case, pair, orc):
    """`pair`: tests/golden/gputest_pair.npz arrays.  Returns a dict of host arrays, the same for every back end."""
    d1, d2 = np.asarray(pair["depth1"]), np.asarray(pair["depth2"])
    rgb1, rgb2 = np.asarray(pair["rgb1"]), np.asarray(pair["rgb2"])
    K = K_FULL
    if case == "small":  # every 5th pixel, cropped to 67 x 49
        d1, d2, rgb1, rgb2 = (np.ascontiguousarray(a[2::5, 1::5][CROP]) for a in (d1, d2, rgb1, rgb2))
        K = (K[0] / 5, K[1] / 5, (K[2] - 1) / 5 - 30, (K[3] - 2) / 5 - 25)
    else:
        assert case == "full"
    verts4, norms4 = helpers.gputest_model_maps(d1, K)          # frame 1 as the model (GPUTest.cpp:69-129), TUM depth / 5000
    depth2 = (d2 // 5).astype(np.uint16)                        # frame 2 as the live depth in mm (GPUTest.cpp:51-57)
    v2 = orc.createVMap(K, (d1 // 5).astype(np.uint16), CUTOFF)  # a second RGBA32F vertex image (the "old" key-frame view): inputs only
    H, W = depth2.shape
    verts4_old = np.zeros((H, W, 4), np.float32)
    okv = ~np.isnan(v2[:H])
    for c in range(3):
        verts4_old[..., c] = np.where(okv, v2[c * H:(c + 1) * H], 0)
    verts4_old = np.roll(verts4_old, 3, axis=1)  # shifted so that the two views disagree about validity in places
    return dict(K=np.array(K, np.float64), depth=depth2, verts4=verts4, norms4=norms4, verts4_old=verts4_old,
                img1=orc.imageBGRToIntensity(helpers.rgba(rgb1)), img2=orc.imageBGRToIntensity(helpers.rgba(rgb2)))


def input_hashes(inp):
    return {"in_" + k: np.array(sha(np.asarray(v))) for k, v in inp.items()}


# ---- the chain --------------------------------------------------------------------------------------------------------------
def chain(ops, inp, feed=None):
    rec = Rec(feed)
    K = [float(v) for v in inp["K"]]
    cam = lambda l: tuple(np.float32(v) / np.float32(1 << l) for v in K)  # CameraModel::operator()(level), types.cuh:190-196
    # -- live half: RGBDOdometry::initICP(filteredDepth) (:118-142)
    depth = [inp["depth"]]
    for l in (1, 2):
        depth.append(rec("depth_L%d" % l, ops.pyrDown(depth[-1])))
    for l in range(3):
        v = rec("vmap_L%d" % l, ops.createVMap(cam(l), depth[l], CUTOFF), True)
        rec("nmap_L%d" % l, ops.createNMap(v), True)
    # -- model half: initICPModel (:168-207)
    v, n = ops.copyMaps(inp["verts4"], inp["norms4"])
    mv, mn = [rec("mvmap_L0", v, True)], [rec("mnmap_L0", n, True)]
    for l in (1, 2):
        mv.append(rec("mvmap_L%d" % l, ops.resizeVMap(mv[-1]), True))
        mn.append(rec("mnmap_L%d" % l, ops.resizeNMap(mn[-1]), True))
    for l in range(3):
        v, n = ops.tranformMaps(mv[l], mn[l], POSE_R, POSE_T)
        rec("tvmap_L%d" % l, v, True)
        rec("tnmap_L%d" % l, n, True)
    # -- populateRGBDData (:209-237): depth from the RGBA32F vertices, float pyramid; intensity pyramid
    dm = [rec("dmap_L0", ops.verticesToDepth(inp["verts4"], MAX_DEPTH_RGB))]
    im = [inp["img1"]]
    im2 = [inp["img2"]]
    for l in (1, 2):
        dm.append(rec("dmap_L%d" % l, ops.pyrDownGaussF(dm[-1])))
        im.append(rec("img1_L%d" % l, ops.pyrDownUcharGauss(im[-1])))
        im2.append(rec("img2_L%d" % l, ops.pyrDownUcharGauss(im2[-1])))
    # -- per level: Sobel pair of the live image (:291), point cloud of the model depth (:412)
    for l in range(3):
        dx, dy = ops.computeDerivativeImages(im2[l])
        rec("dIdx_L%d" % l, dx)
        rec("dIdy_L%d" % l, dy)
        rec("cloud_L%d" % l, ops.projectToPointCloud(dm[l], K, l))
    # -- key-frame operators (KeyFrame.h:46-56, :139, :156-167): vertex-only overloads, 3-plane verticesToDepth
    kv = rec("kvmap", ops.copyMaps(inp["verts4_old"], None), True)
    kv = rec("kvmap_t", ops.tranformMaps(kv, None, POSE_R, POSE_T), True)
    rec("kdepth2d", ops.verticesToDepth2D(kv, CUTOFF))
    d_old = rec("dmap_old", ops.verticesToDepth(inp["verts4_old"], CUTOFF))
    d_new = rec("dmap_new", ops.verticesToDepth(inp["verts4"], CUTOFF))
    # -- NID scores (MutualInformation.cpp:186, :208) at pyramid level 0 and 1, bins as ElasticFusion.h sets them (64 / 500)
    d_old1 = rec("dmap_old_L1", ops.pyrDownGaussF(d_old))
    d_new1 = rec("dmap_new_L1", ops.pyrDownGaussF(d_new))
    nid = []
    for (a, b, c, d, e) in ((im[0], im2[0], d_new, d_old, im2[0]), (im[1], im2[1], d_new1, d_old1, im[1])):
        for bins in (64, 16):
            nid.append(ops.computeNIDImg(a, b, c, d, e, bins))
    for (a, b, c) in ((d_new, d_old, d_old), (d_new1, d_old1, d_new1), (d_old, d_new, d_new)):
        for bins, mx in ((500, 25000.0), (100, 20000.0)):
            nid.append(ops.computeNIDDepth(a, b, c, bins, mx))
    rec("nid", np.array(nid, np.float32))
    return rec.out


# ---- back-end adapters ------------------------------------------------------------------------------------------------------
class OrcOps:
    """The restatement behind the operator names of the product / the reference."""

    def __init__(self, orc):
        from oracle import orc_nid

        self.o, self.nid = orc, orc_nid
        for f in ("pyrDown", "createNMap", "tranformMaps", "copyMaps", "pyrDownGaussF", "pyrDownUcharGauss", "verticesToDepth",
                  "computeDerivativeImages", "projectToPointCloud"):
            setattr(self, f, getattr(orc, f))

    def createVMap(self, cam, depth, cutoff):
        return self.o.createVMap(cam, depth, cutoff)

    def resizeVMap(self, m):
        return self.o.resizeMap(m, False)

    def resizeNMap(self, m):
        return self.o.resizeMap(m, True)

    def verticesToDepth2D(self, vmap, cutoff):
        return self.o.verticesToDepth2D(vmap, cutoff)

    def computeNIDImg(self, a, b, c, d, e, bins):
        return float(self.nid.nid_img(a, b, c, d, e, bins)[0])

    def computeNIDDepth(self, a, b, c, bins, mx):
        return float(self.nid.nid_depth(a, b, c, bins, mx)[0])


class HipOps:
    """The product's operator layer (densemonoslam_amd.odometry.ops, i.e. the C ABI), results downloaded to numpy."""

    def __init__(self, ops):
        self.ops = ops

    def __getattr__(self, name):
        f = getattr(self.ops, name)

        def call(*a):
            r = f(*a)
            if name.startswith("computeNID"):
                return float(r[0])
            if isinstance(r, tuple):
                return tuple(x.download() for x in r)
            return r.download()

        return call
