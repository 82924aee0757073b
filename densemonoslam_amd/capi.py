"""ctypes binding of libdmslam_hip.so (the C ABI declared in include/dmslam.h).

The library is the product: there is no CPU fallback.  Importing this module on a machine
where the shared object has not been built raises ImportError with the build command.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMS_LIB_PATH: another build of the same library (A/B measurements of two builds on one box, scripts/ab_lib.sh); never a fallback
LIB_PATH = os.environ.get("DMS_LIB_PATH") or os.path.join(_HERE, "libdmslam_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "densemonoslam_amd: %s is missing — build it with `make` (hipcc --offload-arch=gfx950) "
        "or `python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback" % LIB_PATH
    )

lib = C.CDLL(LIB_PATH)

DMS_OK = 0
NUM_PYRS = 3
MAX_SENSORS = 8


class Image2D(C.Structure):
    _fields_ = [("data", C.c_void_p), ("pitch", C.c_size_t), ("rows", C.c_int), ("cols", C.c_int)]


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Mat33(C.Structure):
    _fields_ = [("m", C.c_float * 9)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


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
        ("iterations_run", C.c_int * NUM_PYRS),
        ("so3_iterations_run", C.c_int),
        ("rejected_jump", C.c_int),
    ]


DATATERM_DTYPE = np.dtype(
    [("zero_x", "<i2"), ("zero_y", "<i2"), ("one_x", "<i2"), ("one_y", "<i2"), ("diff", "<f4"), ("valid", "<i4")]
)

lib.dms_version.restype = C.c_char_p
lib.dms_last_error.restype = C.c_char_p
lib.dms_reduce_workspace_bytes.restype = C.c_size_t


class DmsError(RuntimeError):
    code = 0  # the DMS_ERR_* status the entry point returned


def check(rc, what=""):
    if rc != DMS_OK:
        e = DmsError("%s failed (%d): %s" % (what, rc, lib.dms_last_error().decode()))
        e.code = rc
        raise e


def mat33(a):
    m = Mat33()
    flat = np.asarray(a, dtype=np.float32).reshape(9)
    for i in range(9):
        m.m[i] = float(flat[i])
    return m


def float3(a):
    a = np.asarray(a, dtype=np.float32).reshape(3)
    return Float3(float(a[0]), float(a[1]), float(a[2]))


lib.dms_stream_sync.argtypes = [C.c_void_p]
lib.dms_memcpy_d2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
lib.dms_stream_create.argtypes = [C.POINTER(C.c_void_p)]
lib.dms_stream_destroy.argtypes = [C.c_void_p]


def mem_info():
    """(free, total) bytes of HBM on the current device."""
    f, t = C.c_size_t(0), C.c_size_t(0)
    check(lib.dms_mem_info(C.byref(f), C.byref(t)), "dms_mem_info")
    return f.value, t.value


def create_stream():
    """A non-blocking HIP stream handle (int) usable as the `stream` argument everywhere."""
    h = C.c_void_p()
    check(lib.dms_stream_create(C.byref(h)), "dms_stream_create")
    return h.value


def destroy_stream(s):
    check(lib.dms_stream_destroy(C.c_void_p(s)), "dms_stream_destroy")


class DeviceBuffer:
    """Owned HBM allocation (dms_device_alloc / dms_device_free)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib.dms_device_alloc(C.byref(p), C.c_size_t(self.nbytes)), "dms_device_alloc")
        self.ptr = p.value

    def free(self):
        if getattr(self, "ptr", None):
            lib.dms_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes, (arr.nbytes, self.nbytes)
        check(lib.dms_memcpy_h2d(C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), C.c_void_p(stream)),
              "dms_memcpy_h2d")
        return self

    def download(self, dtype, shape, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        check(lib.dms_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), C.c_void_p(stream)),
              "dms_memcpy_d2h")
        return out


class DeviceImage:
    """A dense 2-D device image + its dms_image2d view."""

    def __init__(self, rows, cols, dtype, arr=None):
        self.dtype = np.dtype(dtype)
        self.rows, self.cols = int(rows), int(cols)
        self.buf = DeviceBuffer(max(1, self.rows * self.cols * self.dtype.itemsize))
        self.view = Image2D(C.c_void_p(self.buf.ptr), self.cols * self.dtype.itemsize, self.rows, self.cols)
        if arr is not None:
            self.upload(arr)

    @classmethod
    def from_array(cls, arr, dtype=None):
        arr = np.ascontiguousarray(arr if dtype is None else np.asarray(arr, dtype=dtype))
        if arr.dtype.names is None and arr.ndim == 3:  # (rows, cols, k) packed elements, e.g. float3 / rgba8
            rows, cols = arr.shape[0], arr.shape[1]
            elem = np.dtype((arr.dtype, (arr.shape[2],)))
            im = cls(rows, cols, elem)
            im.buf.upload(arr)
            return im
        im = cls(arr.shape[0], arr.shape[1], arr.dtype)
        im.buf.upload(arr)
        return im

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes == self.rows * self.cols * self.dtype.itemsize, (arr.shape, arr.dtype, self.rows, self.cols, self.dtype)
        self.buf.upload(arr)
        return self

    def download(self):
        if self.dtype.subdtype is not None:
            base, shp = self.dtype.subdtype
            return self.buf.download(base, (self.rows, self.cols) + tuple(shp))
        return self.buf.download(self.dtype, (self.rows, self.cols))

    @property
    def ref(self):
        return C.byref(self.view)


def download_view(view, dtype, elems_per_pixel=1):
    """Copy a (possibly pitched) dms_image2d back into a dense numpy array."""
    dtype = np.dtype(dtype)
    row_bytes = view.cols * dtype.itemsize * elems_per_pixel
    raw = np.empty((view.rows, view.pitch), dtype=np.uint8)
    check(lib.dms_memcpy_d2h(raw.ctypes.data_as(C.c_void_p), C.c_void_p(view.data), C.c_size_t(raw.nbytes), None), "dms_memcpy_d2h")
    dense = np.ascontiguousarray(raw[:, :row_bytes]).view(dtype)
    if elems_per_pixel > 1:
        return dense.reshape(view.rows, view.cols, elems_per_pixel)
    return dense.reshape(view.rows, view.cols)


def device_count():
    n = C.c_int(0)
    rc = lib.dms_device_count(C.byref(n))
    return n.value if rc == DMS_OK else 0


# ---- argument types (ctypes would otherwise pass Python floats as double) -----------------
_P = C.c_void_p
_I2 = C.POINTER(Image2D)
_F = C.c_float
_I = C.c_int
_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int)
_M = C.POINTER(Mat33)
_V = C.POINTER(Float3)
_K = C.POINTER(Camera)

lib.dms_icpStep.argtypes = [_M, _V, _I2, _I2, _M, _V, _K, _I2, _I2, _F, _F, _P, C.c_size_t, _FP, _FP, _FP, _I, _I, _P]
lib.dms_rgbStep.argtypes = [_I2, _F, _I2, _F, _F, _I2, _I2, _F, _P, C.c_size_t, _FP, _FP, _I, _I, _P]
lib.dms_so3Step.argtypes = [_I2, _I2, _M, _M, _M, _P, C.c_size_t, _FP, _FP, _FP, _I, _I, _P]
lib.dms_computeRgbResidual.argtypes = [_F, _I2, _I2, _I2, _I2, _I2, _I2, _I2, _P, C.c_size_t, _F, _V, _M, _IP, _IP, _I, _I, _P]
lib.dms_createVMap.argtypes = [_K, _I2, _I2, _F, _P]
lib.dms_createNMap.argtypes = [_I2, _I2, _P]
lib.dms_tranformMaps.argtypes = [_I2, _I2, _M, _V, _I2, _I2, _P]
lib.dms_tranformVMap.argtypes = [_I2, _M, _V, _I2, _P]
lib.dms_copyMaps.argtypes = [_P, _P, _I2, _I2, _P]
lib.dms_copyVMap.argtypes = [_P, _I2, _P]
lib.dms_resizeVMap.argtypes = [_I2, _I2, _P]
lib.dms_resizeNMap.argtypes = [_I2, _I2, _P]
lib.dms_imageBGRToIntensity.argtypes = [_I2, _I2, _P]
lib.dms_verticesToDepth.argtypes = [_P, _I2, _F, _P]
lib.dms_verticesToDepth2D.argtypes = [_I2, _I2, _F, _P]
lib.dms_projectToPointCloud.argtypes = [_I2, _I2, _K, _I, _P]
lib.dms_pyrDown.argtypes = [_I2, _I2, _P]
lib.dms_pyrDownGaussF.argtypes = [_I2, _I2, _P]
lib.dms_pyrDownUcharGauss.argtypes = [_I2, _I2, _P]
lib.dms_computeDerivativeImages.argtypes = [_I2, _I2, _I2, _P]
lib.dms_nid_workspace_bytes.argtypes = [C.c_int]
lib.dms_nid_workspace_bytes.restype = C.c_size_t
lib.dms_computeNIDImg.argtypes = [_I2, _I2, _I2, _I2, _I2, C.c_int, _P, C.c_size_t, C.POINTER(C.c_float), _P]
lib.dms_computeNIDDepth.argtypes = [_I2, _I2, _I2, C.c_int, C.c_float, _P, C.c_size_t, C.POINTER(C.c_float), _P]

lib.dms_fusion_wait_frame_done.argtypes = [_P, _P]
lib.dms_fusion_set_tracker_budget.argtypes = [_P, _I, _I]
lib.dms_fusion_allow_late_frame.argtypes = [_P, _I]
lib.dms_odometry_set_resident_budget.argtypes = [_P, _I, _I]
lib.dms_exact_arith_selftest.argtypes = [_I, _F, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint)]

lib.dms_odometry_create.argtypes = [C.POINTER(_P), _I, _I, _F, _F, _F, _F, _F, _F]
lib.dms_odometry_destroy.argtypes = [_P]
lib.dms_odometry_set_mode.argtypes = [_P, _I, _I, _I, _I]
lib.dms_odometry_set_exec.argtypes = [_P, _I, _I, _I]
lib.dms_odometry_inject_timeout.argtypes = [_P, _I]
lib.dms_odometry_debug_set.argtypes = [_P, C.c_char_p, _I]
lib.dms_odometry_canon_retries.argtypes = [_P, _IP]
lib.dms_odometry_get_mode.argtypes = [_P, _IP, _IP, _IP]
lib.dms_debug_scalar_gn.argtypes = [_P, _P, _F, _P, _P, _P, _F, _F, _F, _F, _I, _P, _P, _P, _P, _P, _P]
lib.dms_debug_scalar_so3.argtypes = [_P, _P, _P, _F, _F, _F, _F, _P, _P, _P]
lib.dms_odometry_initICP_depth.argtypes = [_P, _I2, _F, _P]
lib.dms_odometry_initICP_maps.argtypes = [_P, _P, _P, _F, _P]
lib.dms_odometry_initICPModel.argtypes = [_P, _P, _P, _F, _FP, _P]
lib.dms_odometry_initRGB.argtypes = [_P, _I2, _P]
lib.dms_odometry_initRGBModel.argtypes = [_P, _I2, _P]
lib.dms_odometry_initFirstRGB.argtypes = [_P, _I2, _P]
lib.dms_odometry_initModelFused.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]
lib.dms_odometry_getIncrementalTransformation.argtypes = [_P, _FP, _FP, _I, _F, _I, _I, _I, _I, C.POINTER(TrackResult), _P]
lib.dms_odometry_track_async.argtypes = [_P, _FP, _FP, _I, _F, _I, _I, _I, _I, _P]
lib.dms_odometry_fetch_result.argtypes = [_P, C.POINTER(TrackResult), _P]
lib.dms_odometry_getCovariance.argtypes = [_P, C.POINTER(C.c_double)]
lib.dms_odometry_get_buffer.argtypes = [_P, _I, _I, _I2]
lib.dms_odometry_set_profiling.argtypes = [_P, _I]
lib.dms_odometry_get_kernel_time.argtypes = [_P, C.c_char_p, C.POINTER(C.c_double), _IP]
