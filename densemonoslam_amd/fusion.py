"""Python mirror of the reference's surfel-map API over the C ABI (include/dmslam_fusion.h).

Names follow the reference: `GlobalModel` (Core/src/GlobalModel.h:43-141), `IndexMap`
(Core/src/IndexMap.h:33-205), `ElasticFusion.processFrame` (Core/src/ElasticFusion.h:92-100).
All work happens on the GPU inside libdmslam_hip.so; numpy only crosses at upload / download.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import Camera, DeviceBuffer, DeviceImage, Image2D, TrackResult, check, lib

MAX_SENSORS = capi.MAX_SENSORS
SURFEL_DTYPE = np.dtype([("pos", "<f4", (4,)), ("col", "<f4", (4,)), ("nrm", "<f4", (4,)), ("times", "<f4", (MAX_SENSORS,))])


class PoseBlock(C.Structure):
    _fields_ = [("pose", C.c_float * 16), ("t_inv", C.c_float * 16)]


class IndexMapOut(C.Structure):
    _fields_ = [("index", Image2D), ("vertConf", Image2D), ("colorTime", Image2D), ("normRad", Image2D)]


class PredictOut(C.Structure):
    _fields_ = [("image", Image2D), ("vertex", Image2D), ("normal", Image2D), ("time", Image2D)]


class FusionParams(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("timeDelta", C.c_int), ("confidence", C.c_float), ("depthCut", C.c_float), ("icpWeight", C.c_float),
        ("fastOdom", C.c_int), ("so3", C.c_int), ("frameToFrameRGB", C.c_int), ("pyramid", C.c_int),
        ("hybrid_tracking", C.c_int), ("rgbOnly", C.c_int), ("timeIdx", C.c_int),
        ("maxDepthProcessed", C.c_float), ("model_capacity", C.c_size_t),
        ("pipeline_ingest", C.c_int), ("global_predict", C.c_int),
        ("nid_keyframing", C.c_int), ("nid_threshold", C.c_float), ("nid_depth_lambda", C.c_float), ("nid_bins_img", C.c_int),
        ("nid_bins_depth", C.c_int), ("nid_pyramid_level", C.c_int),
        ("local_loop_closure", C.c_int), ("reloc", C.c_int), ("num_sensors", C.c_int), ("share_projection", C.c_int),
        ("fused_fill_in", C.c_int), ("hybrid_loops", C.c_int),
    ]


class FrameResult(C.Structure):
    _fields_ = [
        ("pose", C.c_float * 16), ("surfels", C.c_uint), ("tick", C.c_int), ("fused", C.c_int), ("fill_in", C.c_int),
        ("weighting", C.c_float), ("nid_score", C.c_float), ("track", TrackResult),
        ("loop_ok", C.c_int), ("loop_constraints", C.c_int), ("loop_icp_error", C.c_float), ("loop_icp_count", C.c_float),
        ("loop_pose", C.c_float * 16), ("loop_cov_diag", C.c_double * 6),
        ("tracking_ok", C.c_int), ("lost", C.c_int),
    ]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_I2 = C.POINTER(Image2D)
_K = C.POINTER(Camera)
lib.dms_model_create.argtypes = [C.POINTER(_P), C.c_size_t, _I, _I]
lib.dms_model_destroy.argtypes = [_P]
lib.dms_model_count.argtypes = [_P, C.POINTER(C.c_uint), _P]
lib.dms_model_capacity.argtypes = [_P]
lib.dms_model_set_num_sensors.argtypes = [_P, _I]
lib.dms_model_set_clean_suffix_min.argtypes = [_P, C.c_size_t]
lib.dms_model_consume.argtypes = [_P, _P, C.POINTER(C.c_float), _P]
lib.dms_model_export_records.argtypes = [_P, _P, C.c_uint, C.POINTER(C.c_uint), _P]
lib.dms_model_consume_records.argtypes = [_P, _P, C.c_uint, C.POINTER(C.c_float), _P]
lib.dms_model_capacity.restype = C.c_size_t
lib.dms_model_download.argtypes = [_P, _P, C.c_uint, C.POINTER(C.c_uint), _P]
lib.dms_model_upload.argtypes = [_P, _P, C.c_uint, _P]
lib.dms_model_download_ref.argtypes = [_P, _P, C.c_uint, C.POINTER(C.c_uint), _P]
lib.dms_model_upload_ref.argtypes = [_P, _P, C.c_uint, _P]
lib.dms_depth_bilateral.argtypes = [_I2, _I2, _F, _P]
lib.dms_depth_metric.argtypes = [_I2, _I2, _F, _P]
lib.dms_model_initialise.argtypes = [_P, _I2, _I2, _I2, _K, _I, _I, _F, _P]
lib.dms_pose_block_set.argtypes = [_P, C.POINTER(C.c_float), _P]
lib.dms_index_map.argtypes = [_P, _P, _K, _I, _I, _F, _I, _P, C.POINTER(IndexMapOut), _P]
lib.dms_splat_predict.argtypes = [_P, _P, _K, _F, _F, _I, _I, _I, _I, _I, _P, C.POINTER(PredictOut), _P]
lib.dms_splat_depth.argtypes = [_P, _P, _K, _F, _F, _I, _I, _I, _I, _P, _I2, _P]
lib.dms_model_fuse.argtypes = [_P, _P, _I, _I, _I2, _I2, _I2, C.POINTER(IndexMapOut), _K, _F, _F, _P, _P]
lib.dms_model_clean.argtypes = [_P, _P, _I, _I, C.POINTER(IndexMapOut), _I2, _K, _F, C.POINTER(C.c_float), _I, _I, _F, _I, _P]
lib.dms_fill_in.argtypes = [C.POINTER(PredictOut), _I2, _I2, _K, _I, _I, C.POINTER(PredictOut), _P]
lib.dms_resize_nn.argtypes = [_I2, _I2, _I, _P]
lib.dms_fusion_default_params.argtypes = [C.POINTER(FusionParams), _I, _I, _F, _F, _F, _F]
lib.dms_fusion_default_params.restype = None
lib.dms_fusion_create.argtypes = [C.POINTER(_P), C.POINTER(FusionParams)]
lib.dms_fusion_destroy.argtypes = [_P]
lib.dms_fusion_process_frame.argtypes = [_P, _P, _I, _P, C.POINTER(C.c_float), _F, _P]
lib.dms_fusion_fetch.argtypes = [_P, C.POINTER(FrameResult), _P]
lib.dms_fusion_inputs_ready.argtypes = [_P, _P]
lib.dms_fusion_inputs_consumed.argtypes = [_P, _P]
lib.dms_fusion_model.argtypes = [_P]
lib.dms_fusion_set_orb_loop.argtypes = [_P, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.dms_fusion_get_global_loop_constraints.argtypes = [_P, C.POINTER(C.c_float), _I, C.POINTER(C.c_int), _P]
lib.dms_fusion_apply_global_loop_begin.argtypes = [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]
lib.dms_fusion_apply_global_loop_end.argtypes = [_P, C.POINTER(C.c_float), _I, _I, _P]
lib.dms_fusion_model.restype = _P
lib.dms_fusion_odometry.argtypes = [_P]
lib.dms_fusion_odometry.restype = _P
lib.dms_fusion_pose_device.argtypes = [_P]
lib.dms_fusion_pose_device.restype = _P
lib.dms_model_sample_graph.argtypes = [_P, _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_int), _P]
lib.dms_fusion_thumbnails.argtypes = [_P, _P, _P]
lib.dms_fusion_frame_block.argtypes = [_P, _P, _P, _P, C.c_int, _P]
lib.dms_fusion_arm_frame_block.argtypes = [_P, _P, _P, _P, C.c_int]
lib.dms_fusion_frame_block_written.argtypes = [_P]
lib.dms_fusion_get_image.argtypes = [_P, _I, _I2]
lib.dms_fusion_process_frame_begin.argtypes = [_P, _P, _I, _P, C.POINTER(C.c_float), _F, _P]
lib.dms_fusion_fetch_loop.argtypes = [_P, C.POINTER(FrameResult), _P]
lib.dms_fusion_process_frame_end.argtypes = [_P, C.POINTER(C.c_float), _I, C.POINTER(C.c_float), _P]
lib.dms_fusion_get_loop_constraints.argtypes = [_P, C.POINTER(C.c_float), _I, C.POINTER(C.c_int)]
lib.dms_fusion_join_map.argtypes = [_P, _P, C.POINTER(C.c_float), _P]
lib.dms_fusion_set_cluster.argtypes = [_P, C.c_int]
lib.dms_fusion_compute_feedback.argtypes = [_P, _P]
lib.dms_fusion_clusters.argtypes = [_P, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.dms_fusion_cluster_model.argtypes = [_P, C.c_int]
lib.dms_fusion_cluster_model.restype = _P
lib.dms_fusion_import_camera.argtypes = [_P, _P, C.POINTER(C.c_float), _I, _P, _I, _P, _P]
lib.dms_relative_transform.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.dms_pose_compose.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.dms_fusion_set_profiling.argtypes = [_P, _I]


class InterMapResult(C.Structure):
    """dms_intermap_result (include/dmslam_fusion.h)"""
    _fields_ = [("accepted", C.c_int), ("cov_ok", C.c_int), ("relativeTransform", C.c_float * 16), ("refinedPose", C.c_float * 16),
                ("cov_diag", C.c_double * 6), ("lastICPError", C.c_float), ("lastICPCount", C.c_float), ("lastRGBError", C.c_float),
                ("lastRGBCount", C.c_float), ("iterations_run", C.c_int * 3), ("so3_iterations_run", C.c_int)]


lib.dms_refframe_create.argtypes = [C.POINTER(_P), _I, _I, _F, _F, _F, _F]
lib.dms_refframe_destroy.argtypes = [_P]
lib.dms_refframe_refine.argtypes = [_P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P, _I, _F, _I, _I, _I, _F, _F, _F,
                                    C.POINTER(InterMapResult), _P]
lib.dms_refframe_odometry.argtypes = [_P]
lib.dms_refframe_odometry.restype = _P
lib.dms_refframe_get_prediction.argtypes = [_P, C.POINTER(PredictOut)]
lib.dms_fusion_get_kernel_time.argtypes = [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]


def _cam(K):
    return Camera(*[float(v) for v in K])  # (fx, fy, cx, cy)


def _img(a, dtype=None):
    if isinstance(a, DeviceImage):
        return a
    return DeviceImage.from_array(a if dtype is None else np.asarray(a, dtype))


def _f16(a):
    a = np.ascontiguousarray(a, np.float32).reshape(16)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def relative_transform(recoveryPose, currPose):
    """relativeTransform = recoveryPose * currPose.inverse() (ReferenceFrame.h:98), the library's fixed float order."""
    a, ap = _f16(recoveryPose)
    b, bp = _f16(currPose)
    o, op = _f16(np.zeros(16, np.float32))
    check(lib.dms_relative_transform(ap, bp, op), "dms_relative_transform")
    return o.reshape(4, 4)


def pose_compose(a, b):
    """a * b for 4 x 4 float matrices in the library's fixed float order (poseGraph / relativeCons re-basing after a merge)."""
    a, ap = _f16(a)
    b, bp = _f16(b)
    o, op = _f16(np.zeros(16, np.float32))
    check(lib.dms_pose_compose(ap, bp, op), "dms_pose_compose")
    return o.reshape(4, 4)


class DevicePose:
    """dms_pose_block in HBM: pose + its inverse (Eigen Matrix4f::inverse on the reference side)."""

    def __init__(self, pose=None):
        self.buf = DeviceBuffer(C.sizeof(PoseBlock))
        self.set(np.eye(4, dtype=np.float32) if pose is None else pose)

    def set(self, pose):
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        check(lib.dms_pose_block_set(C.c_void_p(self.buf.ptr), p.ctypes.data_as(C.POINTER(C.c_float)), None), "dms_pose_block_set")
        return self

    @property
    def ptr(self):
        return C.c_void_p(self.buf.ptr)

    def download(self):
        raw = self.buf.download(np.float32, (32,))
        return raw[:16].reshape(4, 4), raw[16:].reshape(4, 4)


# ---- image-stage operators -------------------------------------------------------------------
def depth_bilateral(depth_u16, maxD):
    d = _img(depth_u16, np.uint16)
    out = DeviceImage(d.rows, d.cols, np.uint16)
    check(lib.dms_depth_bilateral(d.ref, out.ref, maxD, None), "dms_depth_bilateral")
    return out


def depth_metric(depth_u16, maxD):
    d = _img(depth_u16, np.uint16)
    out = DeviceImage(d.rows, d.cols, np.float32)
    check(lib.dms_depth_metric(d.ref, out.ref, maxD, None), "dms_depth_metric")
    return out


def resize_nn(src, drows, dcols):
    s = _img(src)
    elem = s.dtype.itemsize
    out = DeviceImage(drows, dcols, s.dtype)
    check(lib.dms_resize_nn(s.ref, out.ref, elem, None), "dms_resize_nn")
    return out


class PredictionImages:
    """image RGBA8 + vertex/normal RGBA32F + time u16 (IndexMap combined / old framebuffers)."""

    def __init__(self, rows, cols):
        self.image = DeviceImage(rows, cols, np.dtype((np.uint8, (4,))))
        self.vertex = DeviceImage(rows, cols, np.dtype((np.float32, (4,))))
        self.normal = DeviceImage(rows, cols, np.dtype((np.float32, (4,))))
        self.time = DeviceImage(rows, cols, np.uint16)
        self.c = PredictOut(self.image.view, self.vertex.view, self.normal.view, self.time.view)

    def download(self):
        return self.image.download(), self.vertex.download(), self.normal.download(), self.time.download()


def fill_in(existing, depth_filtered_u16, rgba, K, passthrough_geom=False, passthrough_rgb=False):
    d, c = _img(depth_filtered_u16, np.uint16), _img(rgba, np.uint8)
    out = PredictionImages(d.rows, d.cols)
    k = _cam(K)
    check(lib.dms_fill_in(C.byref(existing.c), d.ref, c.ref, C.byref(k), int(passthrough_geom), int(passthrough_rgb), C.byref(out.c), None),
          "dms_fill_in")
    return out


class GlobalModel:
    """Surfel map in HBM (reference class GlobalModel)."""

    def __init__(self, width, height, capacity=1 << 20, handle=None):
        self.width, self.height = width, height
        self._owned = handle is None
        if handle is None:
            h = C.c_void_p()
            check(lib.dms_model_create(C.byref(h), capacity, width, height), "dms_model_create")
            self.h = h
        else:
            self.h = C.c_void_p(handle)
        self.zbuf = DeviceBuffer(width * height * 8)

    def close(self):
        if getattr(self, "h", None) and self._owned:
            lib.dms_model_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setNumSensors(self, n):
        check(lib.dms_model_set_num_sensors(self.h, int(n)), "dms_model_set_num_sensors")

    def setCleanSuffixMin(self, n):
        check(lib.dms_model_set_clean_suffix_min(self.h, int(n)), "dms_model_set_clean_suffix_min")

    def lastCount(self):
        n = C.c_uint(0)
        check(lib.dms_model_count(self.h, C.byref(n), None), "dms_model_count")
        return n.value

    def upload(self, surfels):
        s = np.ascontiguousarray(surfels, SURFEL_DTYPE)
        check(lib.dms_model_upload(self.h, s.ctypes.data_as(C.c_void_p), len(s), None), "dms_model_upload")

    def downloadMap(self):
        n = self.lastCount()
        out = np.zeros(max(n, 1), SURFEL_DTYPE)
        got = C.c_uint(0)
        check(lib.dms_model_download(self.h, out.ctypes.data_as(C.c_void_p), n, C.byref(got), None), "dms_model_download")
        return out[:got.value].copy()

    def downloadMapRef(self):
        """The reference's 15-float / 60-byte records (Shaders/Vertex.cpp:21-50)."""
        n = self.lastCount()
        out = np.zeros((max(n, 1), 15), np.float32)
        got = C.c_uint(0)
        check(lib.dms_model_download_ref(self.h, out.ctypes.data_as(C.c_void_p), n, C.byref(got), None), "dms_model_download_ref")
        return out[:got.value].copy()

    def savePly(self, path, confidenceThreshold, reference_offsets=False):
        """ElasticFusion::savePly for this map (ElasticFusion.cpp:781-885); returns the number of vertices written."""
        n = C.c_uint(0)
        check(lib.dms_model_save_ply(self.h, os.fsencode(path), C.c_float(confidenceThreshold), int(bool(reference_offsets)), C.byref(n)),
              "dms_model_save_ply")
        return n.value

    # -- map merge (GlobalModel::consume) ----------------------------------------------------
    @staticmethod
    def _T(relativeTransform):
        t = np.ascontiguousarray(relativeTransform, np.float32).reshape(16)
        return t, t.ctypes.data_as(C.POINTER(C.c_float))

    def consume(self, other, relativeTransform):
        """Append `other`'s surfels moved by relativeTransform (4x4); `other` is left untouched."""
        keep, tp = self._T(relativeTransform)
        check(lib.dms_model_consume(self.h, other.h, tp, None), "dms_model_consume")
        check(lib.dms_stream_sync(None))

    def sampleGraph(self, sampleRate=5000):
        """Deformation::sampleGraphModel's samples: n x 4 float32 {pos.xyz, init time}, sorted by init time."""
        cap = int(self.lastCount()) // sampleRate + 2
        out = np.zeros((cap, 4), np.float32)
        n = C.c_int(0)
        check(lib.dms_model_sample_graph(self.h, sampleRate, out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n), None),
              "dms_model_sample_graph")
        return out[:n.value].copy()

    def exportRecords(self, max_count=None):
        """(DeviceBuffer of 20-float records, count): the device-side form of downloadMap, for p2p transfers."""
        cap = int(lib.dms_model_capacity(self.h)) if max_count is None else int(max_count)
        n_now = min(self.lastCount(), cap)
        buf = DeviceBuffer(max(n_now, 1) * SURFEL_DTYPE.itemsize)
        n = C.c_uint(0)
        check(lib.dms_model_export_records(self.h, C.c_void_p(buf.ptr), n_now, C.byref(n), None), "dms_model_export_records")
        return buf, n.value

    def consumeRecords(self, records_ptr, count, relativeTransform):
        keep, tp = self._T(relativeTransform)
        check(lib.dms_model_consume_records(self.h, C.c_void_p(records_ptr), count, tp, None), "dms_model_consume_records")
        check(lib.dms_stream_sync(None))

    def initialise(self, rgba, depth_metric, depth_metric_filtered, K, time, timeIdx, maxDepth):
        c, dm, dmf = _img(rgba, np.uint8), _img(depth_metric, np.float32), _img(depth_metric_filtered, np.float32)
        k = _cam(K)
        check(lib.dms_model_initialise(self.h, c.ref, dm.ref, dmf.ref, C.byref(k), time, timeIdx, maxDepth, None), "dms_model_initialise")

    def fuse(self, pose, time, timeIdx, rgba, depth_metric, depth_metric_filtered, indexmap, K, depthCutoff, weighting):
        c, dm, dmf = _img(rgba, np.uint8), _img(depth_metric, np.float32), _img(depth_metric_filtered, np.float32)
        k = _cam(K)
        check(lib.dms_model_fuse(self.h, pose.ptr, time, timeIdx, c.ref, dm.ref, dmf.ref, C.byref(indexmap.c), C.byref(k), depthCutoff,
                                 weighting, None, None), "dms_model_fuse")

    def clean(self, pose, time, timeIdx, indexmap, K, confThreshold, timeDelta, maxDepth, graph=None, depth_synth=None, isFern=False):
        k = _cam(K)
        gptr, gn = None, 0
        if graph is not None and len(graph):
            g = np.ascontiguousarray(graph, np.float32).reshape(-1, 16)
            gptr, gn = g.ctypes.data_as(C.POINTER(C.c_float)), len(g)
        dref = None
        if depth_synth is not None:
            self._ds = _img(depth_synth, np.float32)
            dref = self._ds.ref
        check(lib.dms_model_clean(self.h, pose.ptr, time, timeIdx, C.byref(indexmap.c), dref, C.byref(k), confThreshold, gptr, gn, timeDelta,
                                  maxDepth, int(isFern), None), "dms_model_clean")
        check(lib.dms_stream_sync(None))


class IndexMap:
    """Model rendering into the camera (reference class IndexMap)."""

    def __init__(self, width, height):
        self.width, self.height = width, height
        f4 = np.dtype((np.float32, (4,)))
        self.index = DeviceImage(height, width, np.uint32)
        self.vertConf = DeviceImage(height, width, f4)
        self.colorTime = DeviceImage(height, width, f4)
        self.normRad = DeviceImage(height, width, f4)
        self.c = IndexMapOut(self.index.view, self.vertConf.view, self.colorTime.view, self.normRad.view)
        self.active = PredictionImages(height, width)
        self.inactive = PredictionImages(height, width)
        self.depth = DeviceImage(height, width, np.float32)

    def predictIndices(self, pose, time, timeIdx, model, K, depthCutoff, timeDelta):
        k = _cam(K)
        check(lib.dms_index_map(model.h, pose.ptr, C.byref(k), time, timeIdx, depthCutoff, timeDelta, C.c_void_p(model.zbuf.ptr),
                                C.byref(self.c), None), "dms_index_map")

    def combinedPredict(self, pose, model, K, depthCutoff, confThreshold, time, timeIdx, maxTime, timeDelta, active=True):
        k = _cam(K)
        tgt = self.active if active else self.inactive
        check(lib.dms_splat_predict(model.h, pose.ptr, C.byref(k), depthCutoff, confThreshold, time, timeIdx, maxTime, timeDelta,
                                    1 if active else 0, C.c_void_p(model.zbuf.ptr), C.byref(tgt.c), None), "dms_splat_predict")
        return tgt

    def synthesizeDepth(self, pose, model, K, depthCutoff, confThreshold, time, timeIdx, maxTime, timeDelta):
        k = _cam(K)
        check(lib.dms_splat_depth(model.h, pose.ptr, C.byref(k), depthCutoff, confThreshold, time, timeIdx, maxTime, timeDelta,
                                  C.c_void_p(model.zbuf.ptr), self.depth.ref, None), "dms_splat_depth")
        return self.depth

    def download_index(self):
        return self.index.download(), self.vertConf.download(), self.colorTime.download(), self.normRad.download()


_IMG_TYPES = {0: (np.uint8, 4), 1: (np.uint16, 1), 2: (np.uint16, 1), 3: (np.float32, 1), 4: (np.float32, 1), 5: (np.uint32, 1),
              6: (np.float32, 4), 7: (np.float32, 4), 8: (np.float32, 4), 9: (np.uint8, 4), 10: (np.float32, 4), 11: (np.float32, 4),
              12: (np.uint16, 1), 13: (np.uint8, 4), 14: (np.float32, 4), 15: (np.float32, 4),
              16: (np.uint8, 4), 17: (np.float32, 4), 18: (np.float32, 4), 19: (np.uint16, 1)}


class ElasticFusion:
    """One camera's frame step (ElasticFusion::processFrame with --o --nkf)."""

    def __init__(self, width, height, K, **opts):
        p = FusionParams()
        lib.dms_fusion_default_params(C.byref(p), width, height, K[0], K[1], K[2], K[3])
        for k, v in opts.items():
            if not hasattr(p, k):
                raise TypeError("unknown option %r" % k)
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        check(lib.dms_fusion_create(C.byref(h), C.byref(p)), "dms_fusion_create")
        self.h = h
        self.width, self.height = width, height
        self._rgb = DeviceBuffer(width * height * 4)
        self._depth = DeviceBuffer(width * height * 2)

    def close(self):
        if getattr(self, "h", None):
            lib.dms_fusion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_frame(self, rgb, depth, stream=None):
        # the previous frame's ingest (internal prep stream) may not have read the staging buffers yet
        check(lib.dms_fusion_inputs_consumed(self.h, stream), "dms_fusion_inputs_consumed")
        rgb = np.ascontiguousarray(rgb, np.uint8)
        self._rgb.upload(rgb)
        self._depth.upload(np.ascontiguousarray(depth, np.uint16))
        return rgb.shape[2]

    def inputsReady(self, producer_stream=None):
        """The next frame's input buffers are produced by work enqueued on `producer_stream` so far."""
        check(lib.dms_fusion_inputs_ready(self.h, producer_stream), "dms_fusion_inputs_ready")

    def processFrameAsync(self, rgb_ptr, channels, depth_ptr, inPose=None, weightMultiplier=1.0, stream=None):
        pp = None
        if inPose is not None:
            self._pose = np.ascontiguousarray(inPose, np.float32).reshape(16)
            pp = self._pose.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_fusion_process_frame(self.h, C.c_void_p(rgb_ptr), channels, C.c_void_p(depth_ptr), pp, weightMultiplier, stream),
              "dms_fusion_process_frame")

    def processFrameBegin(self, rgb, depth, inPose=None, weightMultiplier=1.0, stream=None):
        ch = self.upload_frame(rgb, depth, stream)
        pp = None
        if inPose is not None:
            self._pose = np.ascontiguousarray(inPose, np.float32).reshape(16)
            pp = self._pose.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_fusion_process_frame_begin(self.h, C.c_void_p(self._rgb.ptr), ch, C.c_void_p(self._depth.ptr), pp, weightMultiplier,
                                                 stream), "dms_fusion_process_frame_begin")

    # -- after a map merge (ReferenceFrame::consumeReferenceFrame): several cameras, one map ------------------
    def joinMap(self, owner, relativeTransform, stream=None):
        """`owner`'s map consumes this camera's map (moved by relativeTransform); this camera's pose is re-based and its later
        frames track against / fuse into owner's map.  Both contexts in this process, driven from one thread and stream."""
        t, tp = _f16(relativeTransform)
        check(lib.dms_fusion_join_map(self.h, owner.h, tp, stream), "dms_fusion_join_map")
        self._owner = owner  # must outlive this context

    def importCamera(self, owner, pose, tick, last_rgb_ptr, channels, last_depth_ptr, stream=None):
        """This fresh context becomes a camera that arrives from another rank: `pose` (already in owner's frame), `tick`, and the
        last frame it processed (device pointers), from which the live state is rebuilt.  Joins owner's map; fuses nothing."""
        t, tp = _f16(pose)
        check(lib.dms_fusion_import_camera(self.h, owner.h, tp, int(tick), C.c_void_p(last_rgb_ptr), int(channels), C.c_void_p(last_depth_ptr),
                                           stream), "dms_fusion_import_camera")
        self._owner = owner

    def setOrbLoop(self, orbTcwOld, orbTcwNew):
        """Arm the next processFrameBegin with an ORB loop closure (ElasticFusion.cpp:292-350); None, None disarms."""
        if orbTcwOld is None or orbTcwNew is None:
            check(lib.dms_fusion_set_orb_loop(self.h, None, None), "dms_fusion_set_orb_loop")
            return
        self._orb = (np.ascontiguousarray(orbTcwOld, np.float32).reshape(16), np.ascontiguousarray(orbTcwNew, np.float32).reshape(16))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_fusion_set_orb_loop(self.h, fp(self._orb[0]), fp(self._orb[1])), "dms_fusion_set_orb_loop")

    def globalLoopConstraints(self, stream=None):
        """Constraint rows of the last ORB loop closure: n x 7 float32 {orbTcwOld * p, orbTcwNew * p, INACTIVE time}."""
        n = C.c_int(0)
        cap = (self.width // 20) * (self.height // 20)
        out = np.zeros((cap, 7), np.float32)
        check(lib.dms_fusion_get_global_loop_constraints(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n), stream),
              "dms_fusion_get_global_loop_constraints")
        return out[:n.value].copy()

    def applyGlobalLoopBegin(self, orbTcwOld, orbTcwNew, stream=None):
        """ElasticFusion::applyGlobalLoop up to the constraints (ElasticFusion.cpp:1148-1200)"""
        a = np.ascontiguousarray(orbTcwOld, np.float32).reshape(16)
        b = np.ascontiguousarray(orbTcwNew, np.float32).reshape(16)
        fp = lambda v: v.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_fusion_apply_global_loop_begin(self.h, fp(a), fp(b), stream), "dms_fusion_apply_global_loop_begin")

    def applyGlobalLoopEnd(self, graph=None, accepted=False, stream=None):
        """... and from Deformation::constrain's outcome on: predict, predictIndices, clean with the graph (:1222-1239)"""
        gp, nn = None, 0
        if graph is not None and len(graph):
            self._graph = np.ascontiguousarray(graph, np.float32).reshape(-1, 16)
            gp, nn = self._graph.ctypes.data_as(C.POINTER(C.c_float)), len(self._graph)
        check(lib.dms_fusion_apply_global_loop_end(self.h, gp, nn, int(bool(accepted)), stream), "dms_fusion_apply_global_loop_end")

    def fetchLoop(self, stream=None):
        r = FrameResult()
        check(lib.dms_fusion_fetch_loop(self.h, C.byref(r), stream), "dms_fusion_fetch_loop")
        return r

    def processFrameEnd(self, graph=None, newPose=None, stream=None):
        gp, nn, pp = None, 0, None
        if graph is not None and len(graph):
            self._graph = np.ascontiguousarray(graph, np.float32).reshape(-1, 16)
            gp, nn = self._graph.ctypes.data_as(C.POINTER(C.c_float)), len(self._graph)
        if newPose is not None:
            self._newpose = np.ascontiguousarray(newPose, np.float32).reshape(16)
            pp = self._newpose.ctypes.data_as(C.POINTER(C.c_float))
        check(lib.dms_fusion_process_frame_end(self.h, gp, nn, pp, stream), "dms_fusion_process_frame_end")

    def fetch(self, stream=None):
        r = FrameResult()
        check(lib.dms_fusion_fetch(self.h, C.byref(r), stream), "dms_fusion_fetch")
        return r

    def fetch_rc(self, stream=None):
        """(status, result): like fetch, but hands DMS_ERR_CAPACITY / DMS_ERR_TIMEOUT back with the (valid) result."""
        r = FrameResult()
        return int(lib.dms_fusion_fetch(self.h, C.byref(r), stream)), r

    def poseDevice(self):
        """device address of the 4x4 pose the frame step keeps in HBM"""
        return int(lib.dms_fusion_pose_device(self.h))

    def exportPose(self, dst_ptr, stream=None):
        """copy the pose (16 floats) from its place in HBM to device memory at dst_ptr, stream ordered"""
        check(lib.dms_memcpy_d2d_async(C.c_void_p(dst_ptr), C.c_void_p(self.poseDevice()), 64, stream), "dms_memcpy_d2d_async")

    def odometryHandle(self):
        return C.c_void_p(lib.dms_fusion_odometry(self.h))

    def processFrame(self, rgb, depth, inPose=None, weightMultiplier=1.0, cluster=None):
        if cluster is not None:
            self.setCluster(cluster)
        ch = self.upload_frame(rgb, depth)
        self.processFrameAsync(self._rgb.ptr, ch, self._depth.ptr, inPose, weightMultiplier)
        return self.fetch()

    def globalModel(self, cluster=None):
        """The current cluster's map (GlobalModel::model()), or the map of `cluster` (None if it has no buffers)."""
        if cluster is None:
            return GlobalModel(self.width, self.height, handle=lib.dms_fusion_model(self.h))
        h = lib.dms_fusion_cluster_model(self.h, int(cluster))
        return GlobalModel(self.width, self.height, handle=h) if h else None

    # -- per-cluster surfel buffers (processFrame's `cluster`, GlobalModel::isCluster / clusters) ------------
    def setCluster(self, cluster):
        check(lib.dms_fusion_set_cluster(self.h, int(cluster)), "dms_fusion_set_cluster")

    def computeFeedbackBuffers(self, stream=None):
        check(lib.dms_fusion_compute_feedback(self.h, stream), "dms_fusion_compute_feedback")

    def clusters(self):
        """(ids, current id)"""
        ids = (C.c_int * 64)()
        n, cur = C.c_int(0), C.c_int(0)
        check(lib.dms_fusion_clusters(self.h, ids, 64, C.byref(n), C.byref(cur)), "dms_fusion_clusters")
        return [int(ids[i]) for i in range(min(n.value, 64))], int(cur.value)

    def image(self, which):
        v = Image2D()
        check(lib.dms_fusion_get_image(self.h, which, C.byref(v)), "dms_fusion_get_image")
        dt, k = _IMG_TYPES[which]
        return capi.download_view(v, dt, k)

    def imagePtr(self, which):
        """device pointer of one of the context's dense images (dms_fusion_get_image ids; 13 / 14 / 15 = fill-in colour / vertex / normal)"""
        v = Image2D()
        check(lib.dms_fusion_get_image(self.h, which, C.byref(v)), "dms_fusion_get_image")
        return int(v.data)

    def thumbnails(self, block_ptr, stream=None):
        """Pack this frame's W/8 x H/8 fill-in thumbnails [image | vertex | normal] into device memory at block_ptr."""
        check(lib.dms_fusion_thumbnails(self.h, C.c_void_p(block_ptr), stream), "dms_fusion_thumbnails")

    def frameBlock(self, block_ptr, pose_ptr, tick_ptr, tick, stream=None):
        """thumbnails + pose (16 floats) + tick of this frame into device memory, one launch"""
        check(lib.dms_fusion_frame_block(self.h, C.c_void_p(block_ptr), C.c_void_p(pose_ptr), C.c_void_p(tick_ptr), int(tick), stream),
              "dms_fusion_frame_block")

    def armFrameBlock(self, block_ptr, pose_ptr, tick_ptr, tick):
        """the same block written by the NEXT frame's own last kernel (no launch of its own); frameBlockWritten() tells whether it was"""
        check(lib.dms_fusion_arm_frame_block(self.h, C.c_void_p(block_ptr), C.c_void_p(pose_ptr), C.c_void_p(tick_ptr), int(tick)),
              "dms_fusion_arm_frame_block")

    def frameBlockWritten(self):
        return bool(lib.dms_fusion_frame_block_written(self.h))

    def loopConstraints(self):
        """Surface constraints of the last fetched frame's loop candidate: n x 7 float32
        {worldRawPoint, worldModelPoint, source time} (ElasticFusion.cpp:446-474)."""
        n = C.c_int(0)
        cap = (self.width // 20) * (self.height // 20)
        out = np.zeros((cap, 7), np.float32)
        check(lib.dms_fusion_get_loop_constraints(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)),
              "dms_fusion_get_loop_constraints")
        return out[:n.value].copy()

    OPTIONS = {"rgbOnly": 0, "icpWeight": 1, "pyramid": 2, "fastOdom": 3, "so3": 4, "frameToFrameRGB": 5, "confidence": 6, "depthCut": 7}

    def setOption(self, name, value):
        """The reference's run-time setters (ElasticFusion.cpp:1023-1043): setRgbOnly, setIcpWeight, setPyramid, ..."""
        check(lib.dms_fusion_set_option(self.h, self.OPTIONS[name], C.c_double(float(value))), "dms_fusion_set_option")

    def getOption(self, name):
        v = C.c_double(0.0)
        check(lib.dms_fusion_get_option(self.h, self.OPTIONS[name], C.byref(v)), "dms_fusion_get_option")
        return v.value

    def set_profiling(self, on):
        check(lib.dms_fusion_set_profiling(self.h, int(on)))

    def kernel_time(self, name):
        ms, n = C.c_double(0), C.c_int(0)
        check(lib.dms_fusion_get_kernel_time(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class ReferenceFrameRefiner:
    """ReferenceFrame's m_index + m_rgbd and the second half of resolveRelativeTransformationFern (ReferenceFrame.h:66-110):
    dms_refframe_* (include/dmslam_fusion.h)."""

    def __init__(self, width, height, K):
        h = C.c_void_p()
        check(lib.dms_refframe_create(C.byref(h), width, height, float(K[2]), float(K[3]), float(K[0]), float(K[1])), "dms_refframe_create")
        self.h, self.width, self.height = h, width, height

    def close(self):
        if getattr(self, "h", None):
            lib.dms_refframe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def refine(self, owner, recoveryPose, currPose, vertex_ptr, normal_ptr, image_ptr, timeIdx, maxTime, covThresh=1e-05, icpErrThresh=2e-05,
               icpCountThresh=35000, stream=None):
        """owner: the fusion.ElasticFusion whose map was matched (its maxDepthProcessed - truncated to int like the reference's
        `const int depthCutoff` -, confidence and timeDelta are the call's); vertex / normal / image: device pointers to the
        querying camera's dense fill-in textures (RGBA32F, RGBA32F, RGBA8)."""
        r = InterMapResult()
        p = owner.params
        (_ra, rp), (_ca, cp) = _f16(recoveryPose), _f16(currPose)
        check(lib.dms_refframe_refine(self.h, lib.dms_fusion_model(owner.h), rp, cp, C.c_void_p(vertex_ptr),
                                      C.c_void_p(normal_ptr), C.c_void_p(image_ptr), int(p.maxDepthProcessed), float(owner.getOption("confidence")),
                                      int(timeIdx), int(p.timeDelta), int(maxTime), float(covThresh), float(icpErrThresh), float(icpCountThresh),
                                      C.byref(r), stream), "dms_refframe_refine")
        return r

    def prediction(self):
        """the INACTIVE prediction of the last refinement: (image u8 HxWx4, vertex f32 HxWx4, normal f32 HxWx4)"""
        v = PredictOut()
        check(lib.dms_refframe_get_prediction(self.h, C.byref(v)), "dms_refframe_get_prediction")
        return capi.download_view(v.image, np.uint8, 4), capi.download_view(v.vertex, np.float32, 4), capi.download_view(v.normal, np.float32, 4)
