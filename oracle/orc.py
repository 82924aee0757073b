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
        if getattr(self, "h", None):
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


def covariance(lastA):
    a = _c(lastA, np.float64).reshape(36)
    out = np.zeros(36, np.float64)
    lib.orc_covariance(_p(a), _p(out))
    return out.reshape(6, 6)
