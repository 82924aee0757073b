"""ctypes mirror of the fern database (reference class Ferns, Core/src/Ferns.h) — include/dmslam_ferns.h."""
import ctypes as C

import numpy as np

from . import capi
from .capi import DeviceBuffer, Image2D, check, lib

FERN_MAX = 512
BAD_CODE = 255
_P, _I, _F, _I2 = C.c_void_p, C.c_int, C.c_float, C.POINTER(Image2D)


class FernMatch(C.Structure):
    _fields_ = [("closest", _I), ("candidate", _I), ("dissimilarity", _F), ("blockHDAware", _F), ("icp_error", _F), ("icp_count", _F),
                ("photo_error", _F), ("estPose", _F * 16), ("n_constraints", _I)]


lib.dms_ferns_create.argtypes = [C.POINTER(_P), _I, _I, _F, _I, _I, _F, _F, _F, _F, C.c_uint, _I]
lib.dms_ferns_destroy.argtypes = [_P]
lib.dms_ferns_get_table.argtypes = [_P, C.POINTER(_I), C.POINTER(_I)]
lib.dms_ferns_num_frames.argtypes = [_P]
lib.dms_ferns_get_frame.argtypes = [_P, _I, C.POINTER(_F), C.POINTER(_I), C.POINTER(_I), _P]
lib.dms_ferns_encode.argtypes = [_P, _I2, _I2, _I2, _P, _P, _P]
lib.dms_ferns_add_frame.argtypes = [_P, _I2, _I2, _I2, C.POINTER(_F), _I, _F, C.POINTER(_I), _P]
lib.dms_ferns_find_frame.argtypes = [_P, _I2, _I2, _I2, C.POINTER(_F), _I, _I, _I, C.POINTER(FernMatch), C.POINTER(_F), _P]
lib.dms_ferns_find_frame_thumbs.argtypes = [_P, _P, C.POINTER(_F), _I, _I, _I, C.POINTER(FernMatch), C.POINTER(_F), _P]
lib.dms_ferns_search_codes.argtypes = [_P, _P, _P, _I, _I, _P, _P]
lib.dms_ferns_add_frame_async.argtypes = [_P, _I2, _I2, _I2, _P, C.POINTER(_F), _P, _I, _F, _P]
lib.dms_ferns_encode_thumbs.argtypes = [_P, _P, _P, _P, _P]
lib.dms_ferns_publish_block.argtypes = [_P, _P, _P, _P, _P, _I, C.c_float, _P]
lib.dms_ferns_search_blocks.argtypes = [_P, _P, C.c_size_t, _I, _I, C.c_size_t, C.c_size_t, _I, _I, _P, _P, _P]
lib.dms_ferns_search_blocks_hd.argtypes = [_P, _P, C.c_size_t, _I, C.c_size_t, C.c_size_t, _I, _I, _P, _P]
lib.dms_ferns_consume.argtypes = [_P, _P, C.POINTER(_F), _F, C.POINTER(_I), _P]
lib.dms_ferns_record_bytes.argtypes = [_P]
lib.dms_ferns_record_bytes.restype = C.c_size_t
lib.dms_ferns_export_records.argtypes = [_P, _P, _I, C.POINTER(_I), _P]
lib.dms_ferns_consume_records.argtypes = [_P, _P, _I, C.POINTER(_F), _F, C.POINTER(_I), _P]


def _view(ptr, rows, cols, elem):
    return Image2D(C.c_void_p(ptr), cols * elem, rows, cols)


class Ferns:
    """Ferns(n, maxDepth, photoThresh) for a width x height camera with intrinsics K = (fx, fy, cx, cy)."""

    def __init__(self, width, height, K, num=500, maxDepth_mm=3000, photoThresh=115.0, seed=0, capacity=4096):
        self.width, self.height, self.num = int(width), int(height), int(num)
        h = C.c_void_p()
        check(lib.dms_ferns_create(C.byref(h), num, maxDepth_mm, photoThresh, width, height, K[2], K[3], K[0], K[1], seed, capacity),
              "dms_ferns_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib.dms_ferns_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def table(self):
        pos, rgbd = np.zeros((self.num, 2), np.int32), np.zeros((self.num, 4), np.int32)
        check(lib.dms_ferns_get_table(self.h, pos.ctypes.data_as(C.POINTER(_I)), rgbd.ctypes.data_as(C.POINTER(_I))), "dms_ferns_get_table")
        return pos, rgbd

    def status(self):
        """(frames stored, key frames dropped because the database was full) as of the last completed asynchronous add"""
        a, b = C.c_int(0), C.c_int(0)
        lib.dms_ferns_status.argtypes = [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        check(lib.dms_ferns_status(self.h, C.byref(a), C.byref(b)), "dms_ferns_status")
        return a.value, b.value

    def __len__(self):
        return int(lib.dms_ferns_num_frames(self.h))

    def frame(self, i):
        pose, t, g = np.zeros(16, np.float32), _I(0), _I(0)
        codes = np.zeros(self.num, np.uint8)
        check(lib.dms_ferns_get_frame(self.h, i, pose.ctypes.data_as(C.POINTER(_F)), C.byref(t), C.byref(g), codes.ctypes.data_as(_P)),
              "dms_ferns_get_frame")
        return pose.reshape(4, 4), t.value, g.value, codes

    def _tex(self, image, vertex, normal):
        """(device pointer | DeviceBuffer | numpy) x 3 -> three dms_image2d views (uploads numpy inputs)."""
        keep, views = [], []
        for a, elem in ((image, 4), (vertex, 16), (normal, 16)):
            if isinstance(a, np.ndarray):
                b = DeviceBuffer(a.nbytes).upload(np.ascontiguousarray(a))
                keep.append(b)
                a = b.ptr
            elif isinstance(a, DeviceBuffer):
                a = a.ptr
            views.append(_view(int(a), self.height, self.width, elem))
        return keep, views

    def encode(self, image, vertex, normal, stream=None):
        keep, (vi, vv, vn) = self._tex(image, vertex, normal)
        codes, good = DeviceBuffer(FERN_MAX), DeviceBuffer(4)
        check(lib.dms_ferns_encode(self.h, C.byref(vi), C.byref(vv), C.byref(vn), C.c_void_p(codes.ptr), C.c_void_p(good.ptr), stream),
              "dms_ferns_encode")
        return codes.download(np.uint8, (FERN_MAX,), stream)[:self.num], int(good.download(np.int32, (1,), stream)[0])

    def addFrame(self, image, vertex, normal, pose, srcTime, threshold, stream=None):
        keep, (vi, vv, vn) = self._tex(image, vertex, normal)
        p = np.ascontiguousarray(pose, np.float32).reshape(16)
        added = _I(0)
        check(lib.dms_ferns_add_frame(self.h, C.byref(vi), C.byref(vv), C.byref(vn), p.ctypes.data_as(C.POINTER(_F)), srcTime, threshold,
                                      C.byref(added), stream), "dms_ferns_add_frame")
        return bool(added.value)

    def addFrameAsync(self, block_ptr, pose_dev_ptr, srcTime, threshold, stream=None):
        """addFrame of a thumbnail block with the pose read from HBM; no host synchronisation."""
        check(lib.dms_ferns_add_frame_async(self.h, None, None, None, C.c_void_p(block_ptr), None, C.c_void_p(pose_dev_ptr), srcTime, threshold,
                                            stream), "dms_ferns_add_frame_async")

    def publishBlock(self, block_ptr, codes_ptr, good_ptr, pose_dev_ptr, srcTime, threshold, stream=None):
        """encodeThumbs + addFrameAsync of one block with a single encoding pass and no staging copy"""
        check(lib.dms_ferns_publish_block(self.h, C.c_void_p(block_ptr), C.c_void_p(codes_ptr), C.c_void_p(good_ptr), C.c_void_p(pose_dev_ptr),
                                          srcTime, threshold, stream), "dms_ferns_publish_block")

    def encodeThumbs(self, block_ptr, codes_ptr, good_ptr, stream=None):
        check(lib.dms_ferns_encode_thumbs(self.h, C.c_void_p(block_ptr), C.c_void_p(codes_ptr), C.c_void_p(good_ptr), stream),
              "dms_ferns_encode_thumbs")

    def findFrame(self, currPose, vertex, normal, image, time, lost=False, interMap=False, stream=None):
        keep, (vi, vv, vn) = self._tex(image, vertex, normal)
        p = np.ascontiguousarray(currPose, np.float32).reshape(16)
        m, cons = FernMatch(), np.zeros((64, 8), np.float32)
        check(lib.dms_ferns_find_frame(self.h, C.byref(vv), C.byref(vn), C.byref(vi), p.ctypes.data_as(C.POINTER(_F)), time, int(lost),
                                       int(interMap), C.byref(m), cons.ctypes.data_as(C.POINTER(_F)), stream), "dms_ferns_find_frame")
        return m, cons[:m.n_constraints].copy()

    def findFrameThumbs(self, block_ptr, currPose, time, lost=False, interMap=True, stream=None):
        p = np.ascontiguousarray(currPose, np.float32).reshape(16)
        m, cons = FernMatch(), np.zeros((64, 8), np.float32)
        check(lib.dms_ferns_find_frame_thumbs(self.h, C.c_void_p(block_ptr), p.ctypes.data_as(C.POINTER(_F)), time, int(lost), int(interMap),
                                              C.byref(m), cons.ctypes.data_as(C.POINTER(_F)), stream), "dms_ferns_find_frame_thumbs")
        return m, cons[:m.n_constraints].copy()

    def searchCodes(self, codes_ptr, good_ptr, time, interMap, best_ptr, stream=None):
        check(lib.dms_ferns_search_codes(self.h, C.c_void_p(codes_ptr), C.c_void_p(good_ptr), time, int(interMap), C.c_void_p(best_ptr), stream),
              "dms_ferns_search_codes")

    def searchBlocks(self, blocks_ptr, stride, count, skip, codes_offset, good_offset, time, interMap, best_ptr, stream=None, previous_out=None):
        check(lib.dms_ferns_search_blocks(self.h, C.c_void_p(blocks_ptr), stride, count, skip, codes_offset, good_offset, time, int(interMap),
                                          C.c_void_p(best_ptr), C.c_void_p(previous_out) if previous_out else None, stream),
              "dms_ferns_search_blocks")

    def searchBlocksHd(self, blocks_ptr, stride, count, codes_offset, good_offset, time, interMap, hits_ptr, stream=None):
        """search + the blockHDAware test's operands for every block: count x {candidate, dissimilarity bits, valid in both, equal}"""
        check(lib.dms_ferns_search_blocks_hd(self.h, C.c_void_p(blocks_ptr), C.c_size_t(stride), count, C.c_size_t(codes_offset), C.c_size_t(good_offset),
                                             time, int(interMap), C.c_void_p(hits_ptr), stream), "dms_ferns_search_blocks_hd")

    def consume(self, other, relativeTransform, threshold, stream=None):
        T = np.ascontiguousarray(relativeTransform, np.float32).reshape(16)
        added = _I(0)
        check(lib.dms_ferns_consume(self.h, other.h, T.ctypes.data_as(C.POINTER(_F)), threshold, C.byref(added), stream), "dms_ferns_consume")
        return added.value

    # -- the database as records, for a merge across ranks ------------------------------------------------------
    def recordBytes(self):
        return int(lib.dms_ferns_record_bytes(self.h))

    def exportRecords(self, dst_ptr, max_count, stream=None):
        n = _I(0)
        check(lib.dms_ferns_export_records(self.h, C.c_void_p(dst_ptr), int(max_count), C.byref(n), stream), "dms_ferns_export_records")
        return n.value

    def consumeRecords(self, records_ptr, count, relativeTransform, threshold, stream=None):
        T = np.ascontiguousarray(relativeTransform, np.float32).reshape(16)
        added = _I(0)
        check(lib.dms_ferns_consume_records(self.h, C.c_void_p(records_ptr), int(count), T.ctypes.data_as(C.POINTER(_F)), threshold, C.byref(added),
                                            stream), "dms_ferns_consume_records")
        return added.value
