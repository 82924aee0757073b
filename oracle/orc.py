"""ctypes wrapper of the CPU ORACLE (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product path, densemonoslam_amd/, must never do so).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


if not os.path.exists(LIB_PATH):
    build()

lib = C.CDLL(LIB_PATH)

DATATERM_DTYPE = np.dtype(
    [("zero_x", "<i2"), ("zero_y", "<i2"), ("one_x", "<i2"), ("one_y", "<i2"), ("diff", "<f4"), ("valid", "<i4")]
)


class TrackResult(C.Structure):
    _fields_ = [
        ("trans", C.c_float * 3),
        ("rot", C.c_float * 9),
        ("lastICPError", C.c_float),
        ("lastICPCount", C.c_float),
        ("lastRGBError", C.c_float),
        ("lastRGBCount", C.c_float),
        ("lastSO3Error", C.c_float),
        ("lastSO3Count", C.c_float),
        ("lastA", C.c_double * 36),
        ("lastb", C.c_double * 6),
        ("iterations_run", C.c_int * 3),
        ("so3_iterations_run", C.c_int),
        ("rejected_jump", C.c_int),
        ("canon_retries", C.c_int),
        ("trace_len", C.c_int),
        ("trace", (C.c_float * 12) * 160),
    ]


_P = C.c_void_p
_F = C.c_float
_I = C.c_int
lib.orc_qnan.restype = C.c_float
lib.orc_odometry_create.restype = _P
lib.orc_odometry_create.argtypes = [_I, _I, _F, _F, _F, _F, _F, _F]
lib.orc_odometry_destroy.argtypes = [_P]
lib.orc_odometry_set_fused_rows.argtypes = [_P, _I]
lib.orc_set_fused_rows.argtypes = [_I]
lib.orc_odometry_set_sum_mode.argtypes = [_P, _I]
lib.orc_odometry_set_step_hooks.argtypes = [_P, _P, _P, _P, _P]
lib.orc_odometry_set_solve_mode.argtypes = [_P, _I]
lib.orc_odometry_set_exp_bias.argtypes = [_P, _I]
lib.orc_odometry_buffer.restype = _P
lib.orc_odometry_buffer.argtypes = [_P, _I, _I]
lib.orc_odometry_initICP_depth.argtypes = [_P, _P, _F]
lib.orc_odometry_initICP_maps.argtypes = [_P, _P, _P, _F]
lib.orc_odometry_initICPModel.argtypes = [_P, _P, _P, _F, _P]
lib.orc_odometry_initRGB.argtypes = [_P, _P]
lib.orc_odometry_initRGBModel.argtypes = [_P, _P]
lib.orc_odometry_initFirstRGB.argtypes = [_P, _P]
lib.orc_odometry_getIncrementalTransformation.argtypes = [_P, _P, _P, _I, _F, _I, _I, _I, _I, C.POINTER(TrackResult)]

lib.orc_pyrDown.argtypes = [_P, _I, _I, _P]
lib.orc_createVMap.argtypes = [_F, _F, _F, _F, _P, _I, _I, _P, _F]
lib.orc_createNMap.argtypes = [_P, _I, _I, _P]
lib.orc_tranformMaps.argtypes = [_P, _P, _I, _I, _P, _P, _P, _P]
lib.orc_copyMaps.argtypes = [_P, _P, _I, _I, _P, _P]
lib.orc_resizeMap.argtypes = [_P, _I, _I, _P, _I]
lib.orc_pyrDownGaussF.argtypes = [_P, _I, _I, _P]
lib.orc_pyrDownUcharGauss.argtypes = [_P, _I, _I, _P]
lib.orc_verticesToDepth.argtypes = [_P, _I, _I, _P, _F]
lib.orc_imageBGRToIntensity.argtypes = [_P, _I, _I, _P]
lib.orc_computeDerivativeImages.argtypes = [_P, _I, _I, _P, _P]
lib.orc_projectToPointCloud.argtypes = [_P, _I, _I, _P, _F, _F, _F, _F, _I]
lib.orc_icpStep.argtypes = [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _F, _F, _I, _I, _P, _P, _P]
lib.orc_icp_row.argtypes = [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _F, _F, _I, _I, _I, _I, _P]
lib.orc_icp_row.restype = _I
lib.orc_computeRgbResidual.argtypes = [_F, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P, _P]
lib.orc_rgbStep.argtypes = [_P, _F, _P, _F, _F, _P, _P, _F, _I, _I, _P, _P]
lib.orc_so3Step.argtypes = [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P]
lib.orc_covariance.argtypes = [_P, _P]


class KPre(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "ifx", "ify")]


lib.orc_kpre_of.restype = KPre
lib.orc_kpre_of.argtypes = [_F, _F, _F, _F, _I]
lib.orc_scalar_so3_params.argtypes = [_P, C.POINTER(KPre), _P, _P, _P]
lib.orc_scalar_gn_params.argtypes = [_P, C.POINTER(KPre), _P, _P]
lib.orc_scalar_so3_update.argtypes = [_P, _P, _P, _P]
lib.orc_scalar_gn_update.argtypes = [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]
lib.orc_canon_reduce.argtypes = [_I, _P, _P, C.c_long, _P, _P]
lib.orc_canon_reduce.restype = _I


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---- functional wrappers over dense numpy arrays ------------------------------------------
def pyrDown(src):
    src = _c(src, np.uint16)
    dst = np.empty((src.shape[0] // 2, src.shape[1] // 2), np.uint16)
    lib.orc_pyrDown(_p(src), src.shape[0], src.shape[1], _p(dst))
    return dst


def createVMap(cam, depth, cutoff, out=None):
    depth = _c(depth, np.uint16)
    r, c = depth.shape
    vmap = np.zeros((3 * r, c), np.float32) if out is None else out
    lib.orc_createVMap(cam[0], cam[1], cam[2], cam[3], _p(depth), r, c, _p(vmap), cutoff)
    return vmap


def createNMap(vmap, out=None):
    vmap = _c(vmap, np.float32)
    r, c = vmap.shape[0] // 3, vmap.shape[1]
    nmap = np.zeros((3 * r, c), np.float32) if out is None else out
    lib.orc_createNMap(_p(vmap), r, c, _p(nmap))
    return nmap


def tranformMaps(vmap, nmap, R, t):
    vmap = _c(vmap, np.float32).copy()
    r, c = vmap.shape[0] // 3, vmap.shape[1]
    R = _c(R, np.float32).reshape(9)
    t = _c(t, np.float32).reshape(3)
    if nmap is None:
        lib.orc_tranformMaps(_p(vmap), None, r, c, _p(R), _p(t), _p(vmap), None)
        return vmap
    nmap = _c(nmap, np.float32).copy()
    lib.orc_tranformMaps(_p(vmap), _p(nmap), r, c, _p(R), _p(t), _p(vmap), _p(nmap))
    return vmap, nmap


def copyMaps(v4, n4):
    v4 = _c(v4, np.float32)
    r, c = v4.shape[0], v4.shape[1]
    vd = np.zeros((3 * r, c), np.float32)
    if n4 is None:
        lib.orc_copyMaps(_p(v4), None, r, c, _p(vd), None)
        return vd
    n4 = _c(n4, np.float32)
    nd = np.zeros((3 * r, c), np.float32)
    lib.orc_copyMaps(_p(v4), _p(n4), r, c, _p(vd), _p(nd))
    return vd, nd


def resizeMap(m, normalize):
    m = _c(m, np.float32)
    r, c = m.shape[0] // 3, m.shape[1]
    out = np.zeros((3 * (r // 2), c // 2), np.float32)
    lib.orc_resizeMap(_p(m), r, c, _p(out), 1 if normalize else 0)
    return out


def pyrDownGaussF(src):
    src = _c(src, np.float32)
    dst = np.empty((src.shape[0] // 2, src.shape[1] // 2), np.float32)
    lib.orc_pyrDownGaussF(_p(src), src.shape[0], src.shape[1], _p(dst))
    return dst


def pyrDownUcharGauss(src):
    src = _c(src, np.uint8)
    dst = np.empty((src.shape[0] // 2, src.shape[1] // 2), np.uint8)
    lib.orc_pyrDownUcharGauss(_p(src), src.shape[0], src.shape[1], _p(dst))
    return dst


def verticesToDepth(v4, cutoff):
    v4 = _c(v4, np.float32)
    dst = np.empty(v4.shape[:2], np.float32)
    lib.orc_verticesToDepth(_p(v4), v4.shape[0], v4.shape[1], _p(dst), cutoff)
    return dst


def verticesToDepth2D(vmap, cutoff):
    """verticesToDepth2DKernel, cudafuncs.cu:619-630: the z plane of a 3-plane vertex map, NaN where z > cutOff or z <= 0
    (a NaN z fails both tests and is passed through)."""
    vmap = _c(vmap, np.float32)
    r = vmap.shape[0] // 3
    z = vmap[2 * r:3 * r]
    with np.errstate(invalid="ignore"):
        return np.where((z > np.float32(cutoff)) | (z <= 0), np.float32(np.nan), z).astype(np.float32)


def imageBGRToIntensity(rgba):
    rgba = _c(rgba, np.uint8)
    dst = np.empty(rgba.shape[:2], np.uint8)
    lib.orc_imageBGRToIntensity(_p(rgba), rgba.shape[0], rgba.shape[1], _p(dst))
    return dst


def computeDerivativeImages(img):
    img = _c(img, np.uint8)
    dx = np.empty(img.shape, np.int16)
    dy = np.empty(img.shape, np.int16)
    lib.orc_computeDerivativeImages(_p(img), img.shape[0], img.shape[1], _p(dx), _p(dy))
    return dx, dy


def projectToPointCloud(depth, cam, level):
    depth = _c(depth, np.float32)
    cloud = np.empty(depth.shape + (3,), np.float32)
    lib.orc_projectToPointCloud(_p(depth), depth.shape[0], depth.shape[1], _p(cloud), cam[0], cam[1], cam[2], cam[3], level)
    return cloud


def icpStep(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, distThres, angleThres):
    args = [_c(Rcurr, np.float32).reshape(9), _c(tcurr, np.float32).reshape(3), _c(vmap_curr, np.float32), _c(nmap_curr, np.float32),
            _c(Rprev_inv, np.float32).reshape(9), _c(tprev, np.float32).reshape(3)]
    vp, npv = _c(vmap_g_prev, np.float32), _c(nmap_g_prev, np.float32)
    rows, cols = args[2].shape[0] // 3, args[2].shape[1]
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    res = np.zeros(2, np.float32)
    lib.orc_icpStep(_p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(args[4]), _p(args[5]), cam[0], cam[1], cam[2], cam[3],
                    _p(vp), _p(npv), distThres, angleThres, rows, cols, _p(A), _p(b), _p(res))
    return A, b, res


def computeRgbResidual(minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, maxDepthDelta, kt, krkinv):
    dIdx, dIdy = _c(dIdx, np.int16), _c(dIdy, np.int16)
    lastDepth, nextDepth = _c(lastDepth, np.float32), _c(nextDepth, np.float32)
    lastImage, nextImage = _c(lastImage, np.uint8), _c(nextImage, np.uint8)
    kt = _c(kt, np.float32).reshape(3)
    krkinv = _c(krkinv, np.float32).reshape(9)
    rows, cols = nextImage.shape
    corres = np.zeros((rows, cols), DATATERM_DTYPE)
    sigma, count = C.c_int(0), C.c_int(0)
    lib.orc_computeRgbResidual(minScale, _p(dIdx), _p(dIdy), _p(lastDepth), _p(nextDepth), _p(lastImage), _p(nextImage), _p(corres),
                               maxDepthDelta, _p(kt), _p(krkinv), rows, cols, C.byref(sigma), C.byref(count))
    return corres, sigma.value, count.value


def rgbStep(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale):
    corres = np.ascontiguousarray(corres)
    cloud = _c(cloud, np.float32)
    dIdx, dIdy = _c(dIdx, np.int16), _c(dIdy, np.int16)
    rows, cols = dIdx.shape
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    lib.orc_rgbStep(_p(corres), sigma, _p(cloud), fx, fy, _p(dIdx), _p(dIdy), sobelScale, rows, cols, _p(A), _p(b))
    return A, b


def so3Step(lastImage, nextImage, imageBasis, kinv, krlr):
    lastImage, nextImage = _c(lastImage, np.uint8), _c(nextImage, np.uint8)
    ib, ki, kr = (_c(m, np.float32).reshape(9) for m in (imageBasis, kinv, krlr))
    rows, cols = nextImage.shape
    A = np.zeros((3, 3), np.float32)
    b = np.zeros(3, np.float32)
    res = np.zeros(2, np.float32)
    lib.orc_so3Step(_p(lastImage), _p(nextImage), _p(ib), _p(ki), _p(kr), rows, cols, _p(A), _p(b), _p(res))
    return A, b, res


_BUF_TYPES = {0: (np.float32, 3), 1: (np.float32, 3), 2: (np.float32, 3), 3: (np.float32, 3), 4: (np.float32, 1), 5: (np.float32, 1),
              6: (np.uint8, 1), 7: (np.uint8, 1), 8: (np.uint8, 1), 9: (np.int16, 1), 10: (np.int16, 1), 11: (np.float32, -3),
              12: (np.uint16, 1), 13: (DATATERM_DTYPE, 1)}


class Odometry:
    """Oracle mirror of the reference RGBDOdometry (host arrays)."""

    def __init__(self, width, height, cx, cy, fx, fy, distThresh=0.0, angleThresh=0.0):
        self.width, self.height = width, height
        self.h = lib.orc_odometry_create(width, height, cx, cy, fx, fy, distThresh, angleThresh)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # (at interpreter shutdown the module globals may already be gone)
            lib.orc_odometry_destroy(self.h)
            self.h = None

    def initICP(self, filteredDepth, depthCutoff):
        d = _c(filteredDepth, np.uint16)
        lib.orc_odometry_initICP_depth(self.h, _p(d), depthCutoff)

    def initICPMaps(self, verts4, norms4, depthCutoff):
        v, n = _c(verts4, np.float32), _c(norms4, np.float32)
        lib.orc_odometry_initICP_maps(self.h, _p(v), _p(n), depthCutoff)

    def initICPModel(self, verts4, norms4, depthCutoff, pose):
        v, n = _c(verts4, np.float32), _c(norms4, np.float32)
        p = _c(pose, np.float32).reshape(16)
        lib.orc_odometry_initICPModel(self.h, _p(v), _p(n), depthCutoff, _p(p))

    def initRGB(self, rgba):
        a = _c(rgba, np.uint8)
        lib.orc_odometry_initRGB(self.h, _p(a))

    def initRGBModel(self, rgba):
        a = _c(rgba, np.uint8)
        lib.orc_odometry_initRGBModel(self.h, _p(a))

    def initFirstRGB(self, rgba):
        a = _c(rgba, np.uint8)
        lib.orc_odometry_initFirstRGB(self.h, _p(a))

    def setFusedRows(self, on):
        """rows with fused multiply-adds (the product's resident kernels) or with every operation rounded (its operator layer)"""
        lib.orc_odometry_set_fused_rows(self.h, int(bool(on)))

    def setSumMode(self, canonical):
        """cross-pixel sums: canonical order-free sums (default) or fp64 accumulation in loop / thread order (the control)"""
        lib.orc_odometry_set_sum_mode(self.h, int(bool(canonical)))

    def setExpBias(self, bias):
        """test hook: bias of the static exponents of a call's first reductions (the product's "exp_bias")"""
        lib.orc_odometry_set_exp_bias(self.h, int(bias))

    def setStepHooks(self, so3=None, rgbres=None, icp=None, rgb=None):
        """sum mode 0 only: run the host loop around other implementations of the four steps (addresses of functions with the
        signatures of orc_so3Step / orc_computeRgbResidual / orc_icpStep / orc_rgbStep; None = the restatement)"""
        lib.orc_odometry_set_step_hooks(self.h, so3, rgbres, icp, rgb)

    def setSolveMode(self, canonical):
        """scalar section: the product's canonical operation order (default) or the independent Eigen-like restatement (control)"""
        lib.orc_odometry_set_solve_mode(self.h, int(bool(canonical)))

    def getIncrementalTransformation(self, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap=False):
        t = _c(trans, np.float32).reshape(3).copy()
        R = _c(rot, np.float32).reshape(9).copy()
        res = TrackResult()
        lib.orc_odometry_getIncrementalTransformation(self.h, _p(t), _p(R), int(rgbOnly), icpWeight, int(pyramid), int(fastOdom),
                                                      int(so3), int(interMap), C.byref(res))
        return t, R.reshape(3, 3), res

    def buffer(self, which, level):
        dt, k = _BUF_TYPES[which]
        r, c = self.height >> level, self.width >> level
        ptr = lib.orc_odometry_buffer(self.h, which, level)
        dt = np.dtype(dt)
        if k == 3:
            shape = (3 * r, c)
        elif k == -3:
            shape = (r, c, 3)
        else:
            shape = (r, c)
        n = int(np.prod(shape))
        buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape).copy()


def unpack_se3(sums):
    """29 sums (JtJJtrSE3 order) -> A (6x6 float32, symmetric), b (6)"""
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                b[i] = sums[k]
            else:
                A[i, j] = A[j, i] = sums[k]
            k += 1
    return A, b


def scalar_gn_update(sums_icp, sums_rgb, icpWeight, Rprev, tprev, resultRt, cam, next_level):
    """The oracle's restatement of one Gauss-Newton update in the canonical operation order (orc_scalar.c).
    Returns resultRt', A, b, Rcurr, tcurr, krkinv, kt."""
    icp, rgb = sums_icp is not None, sums_rgb is not None
    Ai, bi = unpack_se3(sums_icp) if icp else (np.zeros((6, 6), np.float32), np.zeros(6, np.float32))
    Ar, br = unpack_se3(sums_rgb) if rgb else (np.zeros((6, 6), np.float32), np.zeros(6, np.float32))
    Rp, tp = _c(Rprev, np.float32).reshape(9), _c(tprev, np.float32).reshape(3)
    Rt = _c(resultRt, np.float64).reshape(16).copy()
    A, b = np.zeros(36, np.float64), np.zeros(6, np.float64)
    Rc, tc = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib.orc_scalar_gn_update(_p(Ai), _p(bi), _p(Ar), _p(br), int(icp), int(rgb), icpWeight, _p(Rp), _p(tp), _p(Rt), _p(A), _p(b), _p(Rc), _p(tc))
    k = lib.orc_kpre_of(cam[0], cam[1], cam[2], cam[3], next_level)
    krk, kt = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib.orc_scalar_gn_params(_p(Rt), C.byref(k), _p(krk), _p(kt))
    return Rt, A, b, Rc, tc, krk, kt


def scalar_so3_update(sums, R_lr, resultR, cam):
    """one SO3 update + the next iteration's parameters in the canonical operation order (orc_scalar.c)"""
    jtj = np.zeros(9, np.float32)
    jtr = np.zeros(3, np.float32)
    k = 0
    for i in range(3):
        for j in range(i, 4):
            if j == 3:
                jtr[i] = sums[k]
            else:
                jtj[i * 3 + j] = jtj[j * 3 + i] = sums[k]
            k += 1
    lr = _c(R_lr, np.float32).reshape(9).copy()
    rr = _c(resultR, np.float64).reshape(9).copy()
    lib.orc_scalar_so3_update(_p(jtj), _p(jtr), _p(lr), _p(rr))
    kp = lib.orc_kpre_of(cam[0], cam[1], cam[2], cam[3], 2)
    ib, ki, kr = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(9, np.float32)
    lib.orc_scalar_so3_params(_p(rr), C.byref(kp), _p(ib), _p(ki), _p(kr))
    return lr, rr, ib, ki, kr


def canon_reduce(rows, found, E):
    """canonical order-free sums of per-pixel rows [n][cols] (orc_canon.c); returns sums, E after retries, retries"""
    rows = _c(rows, np.float32)
    n = rows.shape[1] - 1
    fd = _c(found, np.uint8)
    Ei = _c(E, np.int32).copy()
    sums = np.zeros(n * (n + 3) // 2 + 2, np.float32)
    r = lib.orc_canon_reduce(n, _p(rows), _p(fd), rows.shape[0], _p(Ei), _p(sums))
    return sums, Ei, r


def set_threads(n):
    """Threads of the oracle's OpenMP loops (works after libgomp has been initialised by another library)."""
    lib.orc_set_threads(int(n))
    return int(lib.orc_get_threads())


def set_num_sensors(n):
    """Time slots the clean's health test loops over (reference NUM_CAMERAS = 3, Shaders/size.glsl:2)."""
    lib.orc_set_num_sensors(int(n))


def covariance(lastA):
    a = _c(lastA, np.float64).reshape(36)
    out = np.zeros(36, np.float64)
    lib.orc_covariance(_p(a), _p(out))
    return out.reshape(6, 6)


# =============================================================================================
# fusion half (oracle/orc_fusion.c)
# =============================================================================================
MAX_SENSORS = 8
SURFEL_DTYPE = np.dtype([("pos", "<f4", (4,)), ("col", "<f4", (4,)), ("nrm", "<f4", (4,)), ("times", "<f4", (MAX_SENSORS,))])
assert SURFEL_DTYPE.itemsize == 4 * (12 + MAX_SENSORS)

lib.orc_inv4f.argtypes = [_P, _P]
lib.orc_depth_bilateral.argtypes = [_P, _I, _I, _F, _P]
lib.orc_depth_metric.argtypes = [_P, _I, _I, _F, _P]
lib.orc_model_initialise.argtypes = [_P, _P, _P, _I, _I, _F, _F, _F, _F, _I, _I, _F, _P, _I]
lib.orc_model_initialise.restype = _I
lib.orc_index_map.argtypes = [_P, _I, _P, _F, _F, _F, _F, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P]
lib.orc_splat_predict.argtypes = [_P, _I, _P, _F, _F, _F, _F, _I, _I, _F, _F, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]
lib.orc_model_fuse.argtypes = [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _F, _P, C.POINTER(C.c_int)]
lib.orc_model_fuse.restype = _I
lib.orc_model_clean.argtypes = [_P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _I, _I, _F, _I, _P, _I]
lib.orc_model_clean.restype = _I
lib.orc_fill_in.argtypes = [_P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _I, _I, _P, _P, _P]
lib.orc_resize_nn.argtypes = [_P, _I, _I, _P, _I, _I, _I]
lib.orc_dense_enough.argtypes = [_P, _I, _I]
lib.orc_dense_enough.restype = _I
lib.orc_velocity_weight.argtypes = [_P, _P, _F]
lib.orc_velocity_weight.restype = C.c_float


def inv4f(m):
    m = _c(m, np.float32).reshape(16)
    o = np.zeros(16, np.float32)
    lib.orc_inv4f(_p(m), _p(o))
    return o.reshape(4, 4)


def depth_bilateral(depth, maxD):
    d = _c(depth, np.uint16)
    out = np.zeros_like(d)
    lib.orc_depth_bilateral(_p(d), d.shape[0], d.shape[1], maxD, _p(out))
    return out


def depth_metric(depth, maxD):
    d = _c(depth, np.uint16)
    out = np.zeros(d.shape, np.float32)
    lib.orc_depth_metric(_p(d), d.shape[0], d.shape[1], maxD, _p(out))
    return out


def model_initialise(rgba, dm, dmf, cam, time, timeIdx, maxDepth, cap=None):
    rgba, dm, dmf = _c(rgba, np.uint8), _c(dm, np.float32), _c(dmf, np.float32)
    rows, cols = dm.shape
    cap = rows * cols if cap is None else cap
    out = np.zeros(cap, SURFEL_DTYPE)
    n = lib.orc_model_initialise(_p(rgba), _p(dm), _p(dmf), rows, cols, cam[2], cam[3], cam[0], cam[1], time, timeIdx, maxDepth, _p(out), cap)
    assert n >= 0, "raw / filtered feedback buffers have different lengths: the reference would pair mismatched streams"
    return out[:n].copy()


def index_map(model, pose, cam, rows, cols, time, timeIdx, maxDepth, timeDelta):
    model = np.ascontiguousarray(model, SURFEL_DTYPE)
    pose = _c(pose, np.float32).reshape(16)
    index = np.zeros((rows, cols), np.uint32)
    vc, ct, nr = (np.zeros((rows, cols, 4), np.float32) for _ in range(3))
    lib.orc_index_map(_p(model), len(model), _p(pose), cam[2], cam[3], cam[0], cam[1], rows, cols, time, timeIdx, maxDepth, timeDelta,
                      _p(index), _p(vc), _p(ct), _p(nr))
    return index, vc, ct, nr


def splat_predict(model, pose, cam, rows, cols, maxDepth, confThreshold, time, timeIdx, maxTime, timeDelta, active, depth_only=False):
    model = np.ascontiguousarray(model, SURFEL_DTYPE)
    pose = _c(pose, np.float32).reshape(16)
    if depth_only:
        d = np.zeros((rows, cols), np.float32)
        lib.orc_splat_predict(_p(model), len(model), _p(pose), cam[2], cam[3], cam[0], cam[1], rows, cols, maxDepth, confThreshold, time,
                              timeIdx, maxTime, timeDelta, 0, None, None, None, None, _p(d))
        return d
    image = np.zeros((rows, cols, 4), np.uint8)
    vertex = np.zeros((rows, cols, 4), np.float32)
    normal = np.zeros((rows, cols, 4), np.float32)
    timg = np.zeros((rows, cols), np.uint16)
    lib.orc_splat_predict(_p(model), len(model), _p(pose), cam[2], cam[3], cam[0], cam[1], rows, cols, maxDepth, confThreshold, time,
                          timeIdx, maxTime, timeDelta, int(active), _p(image), _p(vertex), _p(normal), _p(timg), None)
    return image, vertex, normal, timg


def model_fuse(model, pose, time, timeIdx, rgba, dr, drf, index, vertConf, normRad, cam, maxDepth, weighting):
    """Returns (updated model, newUnstable list, merged count)."""
    model = np.ascontiguousarray(model, SURFEL_DTYPE).copy()
    pose = _c(pose, np.float32).reshape(16)
    rgba, dr, drf = _c(rgba, np.uint8), _c(dr, np.float32), _c(drf, np.float32)
    index, vertConf, normRad = _c(index, np.uint32), _c(vertConf, np.float32), _c(normRad, np.float32)
    rows, cols = dr.shape
    newU = np.zeros(rows * cols, SURFEL_DTYPE)
    nnew = C.c_int(0)
    merged = lib.orc_model_fuse(_p(model), len(model), _p(pose), time, timeIdx, _p(rgba), _p(dr), _p(drf), _p(index), _p(vertConf),
                                _p(normRad), rows, cols, cam[2], cam[3], cam[0], cam[1], maxDepth, weighting, _p(newU), C.byref(nnew))
    return model, newU[:nnew.value].copy(), merged


def model_clean(model, newU, pose, time, timeIdx, index, vertConf, colorTime, cam, confThreshold, timeDelta, maxDepth, nodes=None,
                depthSynth=None, isFern=0, cap=None):
    model = np.ascontiguousarray(model, SURFEL_DTYPE)
    newU = np.ascontiguousarray(newU, SURFEL_DTYPE)
    pose = _c(pose, np.float32).reshape(16)
    index, vertConf, colorTime = _c(index, np.uint32), _c(vertConf, np.float32), _c(colorTime, np.float32)
    rows, cols = index.shape
    cap = len(model) + len(newU) if cap is None else cap
    out = np.zeros(max(cap, 1), SURFEL_DTYPE)
    nn = 0
    nptr = None
    if nodes is not None and len(nodes):
        nodes = _c(nodes, np.float32).reshape(-1, 16)
        nn, nptr = len(nodes), _p(nodes)
    dptr = None
    if depthSynth is not None:
        depthSynth = _c(depthSynth, np.float32)
        dptr = _p(depthSynth)
    n = lib.orc_model_clean(_p(model), len(model), _p(newU), len(newU), _p(pose), time, timeIdx, _p(index), _p(vertConf), _p(colorTime),
                            dptr, rows, cols, cam[2], cam[3], cam[0], cam[1], confThreshold, nptr, nn, timeDelta, maxDepth, isFern,
                            _p(out), cap)
    return out[:n].copy()


def sample_graph(model, sampleRate):
    """Deformation::sampleGraphModel up to the sort (Deformation.cpp:250-331; sample.vert / sample.geom):
    every sampleRate-th surfel as {pos.xyz, init time}, sorted by init time.  std::sort leaves the
    order of equal times unspecified; the stable order is taken."""
    model = np.ascontiguousarray(model, SURFEL_DTYPE)
    sel = model[::sampleRate]
    rows = np.concatenate([sel["pos"][:, :3], sel["col"][:, 2:3]], axis=1).astype(np.float32)
    return rows[np.argsort(rows[:, 3], kind="stable")]


def model_consume(dst, src, relativeTransform):
    """GlobalModel::consume (GlobalModel.cpp:898-993) with consume.vert: dst ++ transform(src); position
    through the 4x4 (fp32, ((T0 x + T1 y) + T2 z) + T3), normal through its 3x3, everything else kept."""
    out = np.concatenate([dst, src]).copy()
    # the consuming map's own records go through the program too, with the identity (:903-949): a -0 component can come out +0
    for part, T in ((out[:len(dst)], np.eye(4, dtype=np.float32)), (out[len(dst):], np.asarray(relativeTransform, np.float32).reshape(4, 4))):
        p = part["pos"].astype(np.float32).copy()
        n = part["nrm"].astype(np.float32).copy()
        for i in range(3):
            part["pos"][:, i] = ((T[i, 0] * p[:, 0] + T[i, 1] * p[:, 1]) + T[i, 2] * p[:, 2]) + T[i, 3]
            part["nrm"][:, i] = (T[i, 0] * n[:, 0] + T[i, 1] * n[:, 1]) + T[i, 2] * n[:, 2]
    return out


def fill_in(exVertex, exNormal, exImage, depth, rgba, cam, passGeom, passRgb):
    exVertex, exNormal = _c(exVertex, np.float32), _c(exNormal, np.float32)
    exImage, rgba, depth = _c(exImage, np.uint8), _c(rgba, np.uint8), _c(depth, np.uint16)
    rows, cols = depth.shape
    ov, on, oi = np.zeros_like(exVertex), np.zeros_like(exNormal), np.zeros_like(exImage)
    lib.orc_fill_in(_p(exVertex), _p(exNormal), _p(exImage), _p(depth), _p(rgba), rows, cols, cam[2], cam[3], cam[0], cam[1],
                    int(passGeom), int(passRgb), _p(ov), _p(on), _p(oi))
    return ov, on, oi


def resize_nn(src, drows, dcols):
    src = np.ascontiguousarray(src)
    elem = src.dtype.itemsize * (int(np.prod(src.shape[2:])) if src.ndim > 2 else 1)
    dst = np.zeros((drows, dcols) + src.shape[2:], src.dtype)
    lib.orc_resize_nn(_p(src), src.shape[0], src.shape[1], _p(dst), drows, dcols, elem)
    return dst


def dense_enough(image_rgba):
    a = _c(image_rgba, np.uint8)
    return bool(lib.orc_dense_enough(_p(a), a.shape[0], a.shape[1]))


def velocity_weight(currPose, lastPose, weightMultiplier):
    a, b = _c(currPose, np.float32).reshape(16), _c(lastPose, np.float32).reshape(16)
    return float(lib.orc_velocity_weight(_p(a), _p(b), weightMultiplier))
