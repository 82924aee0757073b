"""Python mirror of the reference tracker API over the C ABI.

`RGBDOdometry` keeps the method names and argument meaning of
elasticfusion/Core/src/Utils/RGBDOdometry.h:32-153; the free functions in `ops` keep those of
Core/src/Cuda/cudafuncs.cuh:70-171.  Everything here runs on the GPU through
libdmslam_hip.so; numpy arrays are only uploaded / downloaded at the edges.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import Camera, DeviceBuffer, DeviceImage, Image2D, TrackResult, check, lib


def _img(a):
    """Accept a DeviceImage, or a numpy array (uploaded)."""
    if isinstance(a, DeviceImage):
        return a
    return DeviceImage.from_array(a)


class ops:
    """Operator layer: one static method per reference free function."""

    _ws = None

    @classmethod
    def workspace(cls):
        if cls._ws is None:
            cls._ws = DeviceBuffer(lib.dms_reduce_workspace_bytes())
        return cls._ws

    @staticmethod
    def pyrDown(src):
        src = _img(src)
        dst = DeviceImage(src.rows // 2, src.cols // 2, np.uint16)
        check(lib.dms_pyrDown(src.ref, dst.ref, None), "dms_pyrDown")
        return dst

    @staticmethod
    def createVMap(cam, depth, cutoff):
        depth = _img(depth)
        vmap = DeviceImage(depth.rows * 3, depth.cols, np.float32)
        check(lib.dms_memset(C.c_void_p(vmap.buf.ptr), 0, C.c_size_t(vmap.buf.nbytes), None))
        k = Camera(*[float(v) for v in cam])
        check(lib.dms_createVMap(C.byref(k), depth.ref, vmap.ref, cutoff, None), "dms_createVMap")
        return vmap

    @staticmethod
    def createNMap(vmap):
        vmap = _img(vmap)
        nmap = DeviceImage(vmap.rows, vmap.cols, np.float32)
        check(lib.dms_memset(C.c_void_p(nmap.buf.ptr), 0, C.c_size_t(nmap.buf.nbytes), None))
        check(lib.dms_createNMap(vmap.ref, nmap.ref, None), "dms_createNMap")
        return nmap

    @staticmethod
    def tranformMaps(vmap, nmap, R, t):
        vmap = _img(vmap)
        Rm, tv = capi.mat33(R), capi.float3(t)
        vd = DeviceImage(vmap.rows, vmap.cols, np.float32)
        check(lib.dms_memset(C.c_void_p(vd.buf.ptr), 0, C.c_size_t(vd.buf.nbytes), None))
        if nmap is None:
            check(lib.dms_tranformVMap(vmap.ref, C.byref(Rm), C.byref(tv), vd.ref, None), "dms_tranformVMap")
            return vd
        nmap = _img(nmap)
        nd = DeviceImage(nmap.rows, nmap.cols, np.float32)
        check(lib.dms_memset(C.c_void_p(nd.buf.ptr), 0, C.c_size_t(nd.buf.nbytes), None))
        check(lib.dms_tranformMaps(vmap.ref, nmap.ref, C.byref(Rm), C.byref(tv), vd.ref, nd.ref, None), "dms_tranformMaps")
        return vd, nd

    @staticmethod
    def copyMaps(v4, n4):
        v4 = np.ascontiguousarray(v4, np.float32)
        rows, cols = v4.shape[:2]
        vb = DeviceBuffer(v4.nbytes).upload(v4)
        vd = DeviceImage(rows * 3, cols, np.float32)
        if n4 is None:
            check(lib.dms_copyVMap(C.c_void_p(vb.ptr), vd.ref, None), "dms_copyVMap")
            return vd
        n4 = np.ascontiguousarray(n4, np.float32)
        nb = DeviceBuffer(n4.nbytes).upload(n4)
        nd = DeviceImage(rows * 3, cols, np.float32)
        check(lib.dms_copyMaps(C.c_void_p(vb.ptr), C.c_void_p(nb.ptr), vd.ref, nd.ref, None), "dms_copyMaps")
        return vd, nd

    @staticmethod
    def resizeVMap(m):
        m = _img(m)
        out = DeviceImage((m.rows // 3 // 2) * 3, m.cols // 2, np.float32)
        check(lib.dms_memset(C.c_void_p(out.buf.ptr), 0, C.c_size_t(out.buf.nbytes), None))
        check(lib.dms_resizeVMap(m.ref, out.ref, None), "dms_resizeVMap")
        return out

    @staticmethod
    def resizeNMap(m):
        m = _img(m)
        out = DeviceImage((m.rows // 3 // 2) * 3, m.cols // 2, np.float32)
        check(lib.dms_memset(C.c_void_p(out.buf.ptr), 0, C.c_size_t(out.buf.nbytes), None))
        check(lib.dms_resizeNMap(m.ref, out.ref, None), "dms_resizeNMap")
        return out

    @staticmethod
    def pyrDownGaussF(src):
        src = _img(src)
        dst = DeviceImage(src.rows // 2, src.cols // 2, np.float32)
        check(lib.dms_pyrDownGaussF(src.ref, dst.ref, None), "dms_pyrDownGaussF")
        return dst

    @staticmethod
    def pyrDownUcharGauss(src):
        src = _img(src)
        dst = DeviceImage(src.rows // 2, src.cols // 2, np.uint8)
        check(lib.dms_pyrDownUcharGauss(src.ref, dst.ref, None), "dms_pyrDownUcharGauss")
        return dst

    @staticmethod
    def verticesToDepth(v4, cutoff):
        v4 = np.ascontiguousarray(v4, np.float32)
        vb = DeviceBuffer(v4.nbytes).upload(v4)
        dst = DeviceImage(v4.shape[0], v4.shape[1], np.float32)
        check(lib.dms_verticesToDepth(C.c_void_p(vb.ptr), dst.ref, cutoff, None), "dms_verticesToDepth")
        return dst

    @staticmethod
    def verticesToDepth2D(vmap, cutoff):
        vmap = _img(vmap)
        dst = DeviceImage(vmap.rows // 3, vmap.cols, np.float32)
        check(lib.dms_verticesToDepth2D(vmap.ref, dst.ref, cutoff, None), "dms_verticesToDepth2D")
        return dst

    @staticmethod
    def imageBGRToIntensity(rgba):
        rgba = _img(rgba)
        dst = DeviceImage(rgba.rows, rgba.cols, np.uint8)
        check(lib.dms_imageBGRToIntensity(rgba.ref, dst.ref, None), "dms_imageBGRToIntensity")
        return dst

    @staticmethod
    def computeDerivativeImages(img):
        img = _img(img)
        dx = DeviceImage(img.rows, img.cols, np.int16)
        dy = DeviceImage(img.rows, img.cols, np.int16)
        check(lib.dms_computeDerivativeImages(img.ref, dx.ref, dy.ref, None), "dms_computeDerivativeImages")
        return dx, dy

    @staticmethod
    def computeNIDImg(img_kf, img_kf_old, dmap_kf, dmap_kf_old, img_curr, num_bins=64):
        """nid of the intensity images (reference computeNIDImg); also returns the histogram (live bin, key-frame bin)."""
        a, b, c, d, e = _img(img_kf), _img(img_kf_old), _img(dmap_kf), _img(dmap_kf_old), _img(img_curr)
        n = lib.dms_nid_workspace_bytes(num_bins)
        ws = DeviceBuffer(n)
        out = C.c_float(0)
        check(lib.dms_computeNIDImg(a.ref, b.ref, c.ref, d.ref, e.ref, num_bins, C.c_void_p(ws.ptr), n, C.byref(out), None), "dms_computeNIDImg")
        hist = ws.download(np.uint32, (num_bins, num_bins))
        return out.value, hist

    @staticmethod
    def computeNIDDepth(dmap_kf, dmap_kf_old, dmap_curr, num_bins=500, max_depth_mm=25000.0):
        a, b, c = _img(dmap_kf), _img(dmap_kf_old), _img(dmap_curr)
        n = lib.dms_nid_workspace_bytes(num_bins)
        ws = DeviceBuffer(n)
        out = C.c_float(0)
        check(lib.dms_computeNIDDepth(a.ref, b.ref, c.ref, num_bins, max_depth_mm, C.c_void_p(ws.ptr), n, C.byref(out), None), "dms_computeNIDDepth")
        hist = ws.download(np.uint32, (num_bins, num_bins))
        return out.value, hist

    @staticmethod
    def projectToPointCloud(depth, cam, level):
        depth = _img(depth)
        cloud = DeviceImage(depth.rows, depth.cols, np.dtype((np.float32, (3,))))
        k = Camera(*[float(v) for v in cam])
        check(lib.dms_projectToPointCloud(depth.ref, cloud.ref, C.byref(k), level, None), "dms_projectToPointCloud")
        return cloud

    @classmethod
    def icpStep(cls, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, distThres, angleThres,
                threads=0, blocks=0):
        vc, nc, vp, npv = _img(vmap_curr), _img(nmap_curr), _img(vmap_g_prev), _img(nmap_g_prev)
        Rc, tc, Rp, tp = capi.mat33(Rcurr), capi.float3(tcurr), capi.mat33(Rprev_inv), capi.float3(tprev)
        k = Camera(*[float(v) for v in cam])
        A = np.zeros((6, 6), np.float32)
        b = np.zeros(6, np.float32)
        res = np.zeros(2, np.float32)
        ws = cls.workspace()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_icpStep(C.byref(Rc), C.byref(tc), vc.ref, nc.ref, C.byref(Rp), C.byref(tp), C.byref(k), vp.ref, npv.ref,
                              distThres, angleThres, C.c_void_p(ws.ptr), ws.nbytes, fp(A), fp(b), fp(res), threads, blocks, None),
              "dms_icpStep")
        return A, b, res

    @classmethod
    def computeRgbResidual(cls, minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, maxDepthDelta, kt, krkinv,
                           threads=0, blocks=0):
        dIdx, dIdy, lastDepth, nextDepth = _img(dIdx), _img(dIdy), _img(lastDepth), _img(nextDepth)
        lastImage, nextImage = _img(lastImage), _img(nextImage)
        corres = DeviceImage(nextImage.rows, nextImage.cols, capi.DATATERM_DTYPE)
        ktv, H = capi.float3(kt), capi.mat33(krkinv)
        sigma, count = C.c_int(0), C.c_int(0)
        ws = cls.workspace()
        check(lib.dms_computeRgbResidual(minScale, dIdx.ref, dIdy.ref, lastDepth.ref, nextDepth.ref, lastImage.ref, nextImage.ref,
                                         corres.ref, C.c_void_p(ws.ptr), ws.nbytes, maxDepthDelta, C.byref(ktv), C.byref(H),
                                         C.byref(sigma), C.byref(count), threads, blocks, None), "dms_computeRgbResidual")
        return corres, sigma.value, count.value

    @classmethod
    def rgbStep(cls, corres, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale, threads=0, blocks=0):
        corres, cloud, dIdx, dIdy = _img(corres), _img(cloud), _img(dIdx), _img(dIdy)
        A = np.zeros((6, 6), np.float32)
        b = np.zeros(6, np.float32)
        ws = cls.workspace()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_rgbStep(corres.ref, sigma, cloud.ref, fx, fy, dIdx.ref, dIdy.ref, sobelScale, C.c_void_p(ws.ptr), ws.nbytes,
                              fp(A), fp(b), threads, blocks, None), "dms_rgbStep")
        return A, b

    @classmethod
    def so3Step(cls, lastImage, nextImage, imageBasis, kinv, krlr, threads=0, blocks=0):
        lastImage, nextImage = _img(lastImage), _img(nextImage)
        ib, ki, kr = capi.mat33(imageBasis), capi.mat33(kinv), capi.mat33(krlr)
        A = np.zeros((3, 3), np.float32)
        b = np.zeros(3, np.float32)
        res = np.zeros(2, np.float32)
        ws = cls.workspace()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_so3Step(lastImage.ref, nextImage.ref, C.byref(ib), C.byref(ki), C.byref(kr), C.c_void_p(ws.ptr), ws.nbytes,
                              fp(A), fp(b), fp(res), threads, blocks, None), "dms_so3Step")
        return A, b, res


_BUF_TYPES = {0: (np.float32, 1), 1: (np.float32, 1), 2: (np.float32, 1), 3: (np.float32, 1), 4: (np.float32, 1), 5: (np.float32, 1),
              6: (np.uint8, 1), 7: (np.uint8, 1), 8: (np.uint8, 1), 9: (np.int16, 1), 10: (np.int16, 1), 11: (np.float32, 3),
              12: (np.uint16, 1), 13: (capi.DATATERM_DTYPE, 1), 14: (np.uint8, 1)}


class RGBDOdometry:
    """Device-resident tracker (reference class RGBDOdometry, Utils/RGBDOdometry.h:32-153).

    Textures of the reference (`GPUTexture*`) are replaced by device images:
      filteredDepth  -> u16 H×W (mm);  rgb -> RGBA8 H×W;
      predictedVertices / predictedNormals -> dense RGBA32F H×W (DeviceBuffer or numpy array).
    """

    def __init__(self, width, height, cx, cy, fx, fy, distThresh=0.0, angleThresh=0.0, stream=None):
        self.width, self.height = int(width), int(height)
        self.stream = stream
        h = C.c_void_p()
        check(lib.dms_odometry_create(C.byref(h), width, height, cx, cy, fx, fy, distThresh, angleThresh), "dms_odometry_create")
        self.h = h
        self.last = None

    def close(self):
        if getattr(self, "h", None):
            lib.dms_odometry_destroy(self.h)
            self.h = None

    def setMode(self, resident=-1, fp64_sums=-1, early_exit=-1, atomic_reduce=-1, coarse_launch=-1):
        """Execution switches of this handle (dms_odometry_set_exec); -1 keeps the current setting.  coarse_launch: SO3 + level 2 + level 1
        in one resident launch (round 6; same bits).  (fp64_sums / atomic_reduce: accepted and ignored, as the library has since round 3.)"""
        check(lib.dms_odometry_set_exec(self.h, int(resident), int(early_exit), int(coarse_launch)), "dms_odometry_set_exec")

    def getMode(self):
        """(resident, max_resident_blocks, fell_back) of this handle"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib.dms_odometry_get_mode(self.h, C.byref(a), C.byref(b), C.byref(c)), "dms_odometry_get_mode")
        return bool(a.value), b.value, bool(c.value)

    def setExpBias(self, bias):
        """test hook: bias of the static exponents of a call's first reductions (csrc/canon.hpp)"""
        check(lib.dms_odometry_debug_set(self.h, b"exp_bias", int(bias)), "dms_odometry_debug_set")

    def canonRetries(self):
        """reductions of the last fetched call that were repeated on a coarser grid"""
        n = C.c_int(0)
        check(lib.dms_odometry_canon_retries(self.h, C.byref(n)), "dms_odometry_canon_retries")
        return n.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _dense(a):
        """RGBA32F map: numpy (H,W,4) is uploaded; DeviceBuffer / int pointer passed through."""
        if isinstance(a, DeviceBuffer):
            return a, a.ptr
        if isinstance(a, DeviceImage):
            return a, a.buf.ptr
        if isinstance(a, int):
            return None, a
        arr = np.ascontiguousarray(a, np.float32)
        buf = DeviceBuffer(arr.nbytes).upload(arr)
        return buf, buf.ptr

    @staticmethod
    def _rgba(a):
        if isinstance(a, DeviceImage):
            return a
        arr = np.ascontiguousarray(a, np.uint8)
        if arr.ndim == 3 and arr.shape[2] == 3:  # reference uploads RGB8 into an RGBA texture (ElasticFusion.cpp:111)
            arr = np.concatenate([arr, np.full(arr.shape[:2] + (1,), 255, np.uint8)], axis=2)
        return DeviceImage.from_array(arr)

    # -- reference API ---------------------------------------------------------------------
    def initICP(self, filteredDepth, depthCutoff):
        d = _img(filteredDepth)
        check(lib.dms_odometry_initICP_depth(self.h, d.ref, depthCutoff, self.stream), "initICP")

    def initICPMaps(self, predictedVertices, predictedNormals, depthCutoff):
        kv, pv = self._dense(predictedVertices)
        kn, pn = self._dense(predictedNormals)
        check(lib.dms_odometry_initICP_maps(self.h, C.c_void_p(pv), C.c_void_p(pn), depthCutoff, self.stream), "initICP(maps)")
        check(lib.dms_stream_sync(self.stream))

    def initICPModel(self, predictedVertices, predictedNormals, depthCutoff, modelPose):
        kv, pv = self._dense(predictedVertices)
        kn, pn = self._dense(predictedNormals)
        pose = np.ascontiguousarray(modelPose, np.float32).reshape(16)
        check(lib.dms_odometry_initICPModel(self.h, C.c_void_p(pv), C.c_void_p(pn), depthCutoff,
                                            pose.ctypes.data_as(C.POINTER(C.c_float)), self.stream), "initICPModel")
        check(lib.dms_stream_sync(self.stream))

    def initRGB(self, rgb):
        check(lib.dms_odometry_initRGB(self.h, self._rgba(rgb).ref, self.stream), "initRGB")
        check(lib.dms_stream_sync(self.stream))

    def initRGBModel(self, rgb):
        check(lib.dms_odometry_initRGBModel(self.h, self._rgba(rgb).ref, self.stream), "initRGBModel")
        check(lib.dms_stream_sync(self.stream))

    def initModelFused(self, vertA, normA, rgbaA, vertB, normB, rgbaB, use_b, force_b_image, modelPose):
        """Frame-step form of initICPModel + initRGBModel: source chosen by a device flag, pose read from HBM."""
        keep = [self._dense(a) for a in (vertA, normA, vertB, normB)]
        ia, ib = self._rgba(rgbaA), self._rgba(rgbaB)
        flag = DeviceBuffer(4).upload(np.array([1 if use_b else 0], np.int32))
        pose = DeviceBuffer(64).upload(np.ascontiguousarray(modelPose, np.float32).reshape(16))
        check(lib.dms_odometry_initModelFused(self.h, C.c_void_p(keep[0][1]), C.c_void_p(keep[1][1]), C.c_void_p(ia.buf.ptr),
                                              C.c_void_p(keep[2][1]), C.c_void_p(keep[3][1]), C.c_void_p(ib.buf.ptr), C.c_void_p(flag.ptr),
                                              1 if force_b_image else 0, C.c_void_p(pose.ptr), self.stream), "initModelFused")
        check(lib.dms_stream_sync(self.stream))

    def initFirstRGB(self, rgb):
        check(lib.dms_odometry_initFirstRGB(self.h, self._rgba(rgb).ref, self.stream), "initFirstRGB")
        check(lib.dms_stream_sync(self.stream))

    def getIncrementalTransformation(self, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, interMap=False):
        t = np.ascontiguousarray(trans, np.float32).reshape(3).copy()
        R = np.ascontiguousarray(rot, np.float32).reshape(9).copy()
        res = TrackResult()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_odometry_getIncrementalTransformation(self.h, fp(t), fp(R), int(rgbOnly), icpWeight, int(pyramid), int(fastOdom),
                                                            int(so3), int(interMap), C.byref(res), self.stream),
              "getIncrementalTransformation")
        self.last = res
        return t, R.reshape(3, 3), res

    def getCovariance(self):
        cov = np.zeros(36, np.float64)
        check(lib.dms_odometry_getCovariance(self.h, cov.ctypes.data_as(C.POINTER(C.c_double))), "getCovariance")
        return cov.reshape(6, 6)

    # -- introspection ---------------------------------------------------------------------
    def buffer(self, which, level):
        v = Image2D()
        check(lib.dms_odometry_get_buffer(self.h, which, level, C.byref(v)), "get_buffer")
        dt, k = _BUF_TYPES[which]
        return capi.download_view(v, dt, k)

    def set_profiling(self, on):
        check(lib.dms_odometry_set_profiling(self.h, int(on)))

    def kernel_time(self, name):
        ms, n = C.c_double(0), C.c_int(0)
        check(lib.dms_odometry_get_kernel_time(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
