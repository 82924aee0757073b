"""Test infrastructure: ctypes binding of oracle/_ref/libref_cudafuncs.so = the REFERENCE's own pyramid / preparation
operators and NID scores (elasticfusion/Core/src/Cuda/cudafuncs.cu minus its texture sampler) built for gfx950 by
oracle/ref_build.sh.  Same function names and argument order as the product's operator layer
(densemonoslam_amd.odometry.ops) and the restatement (tests/ref_cases_cf.py OrcOps) so one case runner drives all three;
everything here takes and returns dense numpy arrays.

Needs a GPU (the reference's kernels run on it).  Used by tests/golden/make_ref_cudafuncs_golden.py (which records the
reference's outputs as committed fixtures) and by the `-m gpu` tests that compare the HIP path with the reference live.
Never imported by the product."""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_cudafuncs.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref_cudafuncs.so is missing: run oracle/ref_build.sh where /root/reference exists")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(v):
    return C.c_float(float(v))


def _ok(rc, what):
    assert rc == 0, (what, rc)


def pyrDown(src):
    src = _c(src, np.uint16)
    dst = np.zeros((src.shape[0] // 2, src.shape[1] // 2), np.uint16)
    _ok(lib().ref_cf_pyrDown(_p(src), src.shape[0], src.shape[1], _p(dst)), "pyrDown")
    return dst


def createVMap(cam, depth, cutoff):
    depth = _c(depth, np.uint16)
    r, c = depth.shape
    vmap = np.zeros((3 * r, c), np.float32)
    _ok(lib().ref_cf_createVMap(_f(cam[0]), _f(cam[1]), _f(cam[2]), _f(cam[3]), _p(depth), r, c, _p(vmap), _f(cutoff)), "createVMap")
    return vmap


def createNMap(vmap):
    vmap = _c(vmap, np.float32)
    r, c = vmap.shape[0] // 3, vmap.shape[1]
    nmap = np.zeros((3 * r, c), np.float32)
    _ok(lib().ref_cf_createNMap(_p(vmap), r, c, _p(nmap)), "createNMap")
    return nmap


def tranformMaps(vmap, nmap, R, t):
    vmap = _c(vmap, np.float32).copy()
    r, c = vmap.shape[0] // 3, vmap.shape[1]
    R = _c(R, np.float32).reshape(9)
    t = _c(t, np.float32).reshape(3)
    if nmap is None:
        _ok(lib().ref_cf_tranformMaps(_p(vmap), None, r, c, _p(R), _p(t)), "tranformMaps")
        return vmap
    nmap = _c(nmap, np.float32).copy()
    _ok(lib().ref_cf_tranformMaps(_p(vmap), _p(nmap), r, c, _p(R), _p(t)), "tranformMaps")
    return vmap, nmap


def copyMaps(v4, n4):
    v4 = _c(v4, np.float32)
    r, c = v4.shape[:2]
    vd = np.zeros((3 * r, c), np.float32)
    if n4 is None:
        _ok(lib().ref_cf_copyMaps(_p(v4), None, r, c, _p(vd), None), "copyMaps")
        return vd
    n4 = _c(n4, np.float32)
    nd = np.zeros((3 * r, c), np.float32)
    _ok(lib().ref_cf_copyMaps(_p(v4), _p(n4), r, c, _p(vd), _p(nd)), "copyMaps")
    return vd, nd


def resizeMap(m, normalize):
    m = _c(m, np.float32)
    r, c = m.shape[0] // 3, m.shape[1]
    out = np.zeros((3 * (r // 2), c // 2), np.float32)
    _ok(lib().ref_cf_resizeMap(_p(m), r, c, _p(out), 1 if normalize else 0), "resizeMap")
    return out


def resizeVMap(m):
    return resizeMap(m, False)


def resizeNMap(m):
    return resizeMap(m, True)


def pyrDownGaussF(src):
    src = _c(src, np.float32)
    dst = np.zeros((src.shape[0] // 2, src.shape[1] // 2), np.float32)
    _ok(lib().ref_cf_pyrDownGaussF(_p(src), src.shape[0], src.shape[1], _p(dst)), "pyrDownGaussF")
    return dst


def pyrDownUcharGauss(src):
    src = _c(src, np.uint8)
    dst = np.zeros((src.shape[0] // 2, src.shape[1] // 2), np.uint8)
    _ok(lib().ref_cf_pyrDownUcharGauss(_p(src), src.shape[0], src.shape[1], _p(dst)), "pyrDownUcharGauss")
    return dst


def verticesToDepth(v4, cutoff):
    v4 = _c(v4, np.float32)
    dst = np.zeros(v4.shape[:2], np.float32)
    _ok(lib().ref_cf_verticesToDepth(_p(v4), v4.shape[0], v4.shape[1], _p(dst), _f(cutoff)), "verticesToDepth")
    return dst


def verticesToDepth2D(vmap, cutoff):
    vmap = _c(vmap, np.float32)
    r, c = vmap.shape[0] // 3, vmap.shape[1]
    dst = np.zeros((r, c), np.float32)
    _ok(lib().ref_cf_verticesToDepth2D(_p(vmap), r, c, _p(dst), _f(cutoff)), "verticesToDepth2D")
    return dst


def computeDerivativeImages(img):
    img = _c(img, np.uint8)
    dx = np.zeros(img.shape, np.int16)
    dy = np.zeros(img.shape, np.int16)
    _ok(lib().ref_cf_computeDerivativeImages(_p(img), img.shape[0], img.shape[1], _p(dx), _p(dy)), "computeDerivativeImages")
    return dx, dy


def projectToPointCloud(depth, cam, level):
    depth = _c(depth, np.float32)
    cloud = np.zeros(depth.shape + (3,), np.float32)
    _ok(lib().ref_cf_projectToPointCloud(_p(depth), depth.shape[0], depth.shape[1], _p(cloud), _f(cam[0]), _f(cam[1]), _f(cam[2]),
                                         _f(cam[3]), int(level)), "projectToPointCloud")
    return cloud


def computeNIDImg(img_kf, img_kf_old, dmap_kf, dmap_kf_old, img_curr, num_bins=64):
    a, b, e = _c(img_kf, np.uint8), _c(img_kf_old, np.uint8), _c(img_curr, np.uint8)
    c, d = _c(dmap_kf, np.float32), _c(dmap_kf_old, np.float32)
    out = C.c_float(0)
    _ok(lib().ref_cf_computeNIDImg(_p(a), _p(b), _p(c), _p(d), _p(e), a.shape[0], a.shape[1], int(num_bins), C.byref(out)), "computeNIDImg")
    return out.value


def computeNIDDepth(dmap_kf, dmap_kf_old, dmap_curr, num_bins=500, max_depth_mm=25000.0):
    a, b, c = _c(dmap_kf, np.float32), _c(dmap_kf_old, np.float32), _c(dmap_curr, np.float32)
    out = C.c_float(0)
    _ok(lib().ref_cf_computeNIDDepth(_p(a), _p(b), _p(c), a.shape[0], a.shape[1], int(num_bins), _f(max_depth_mm), C.byref(out)),
        "computeNIDDepth")
    return out.value
