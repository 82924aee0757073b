"""ctypes mirror of include/dmslam_io.h: eflcm.Frame messages, LCM event logs and .klg logs
(the reference's LcmHandler / RawLcmLogReader / RawLogReader).  Host-side only."""
import ctypes as C

import numpy as np

from .capi import check, lib

DMS_EOF = 1


class FrameMsg(C.Structure):
    _fields_ = [("trackOnly", C.c_int), ("compressed", C.c_int), ("last", C.c_int), ("depthSize", C.c_int32), ("imageSize", C.c_int32),
                ("depth", C.c_void_p), ("image", C.c_void_p), ("timestamp", C.c_int64), ("frameNumber", C.c_int32),
                ("senderName", C.c_char * 128)]


_P = C.c_void_p
DMS_ERR_UNSUPPORTED, DMS_ERR_FORMAT = -7, -8  # include/dmslam_io.h
lib.dms_eflcm_frame_encoded_size.argtypes = [C.POINTER(FrameMsg)]
lib.dms_eflcm_frame_encoded_size.restype = C.c_size_t
lib.dms_eflcm_frame_encode.argtypes = [C.POINTER(FrameMsg), _P, C.c_size_t, C.POINTER(C.c_size_t)]
lib.dms_eflcm_frame_decode.argtypes = [_P, C.c_size_t, C.POINTER(FrameMsg)]
lib.dms_frame_unpack.argtypes = [C.POINTER(FrameMsg), C.c_int, C.c_int, C.c_int, _P, _P]
lib.dms_lcmlog_open.argtypes = [C.POINTER(_P), C.c_char_p]
lib.dms_lcmlog_next.argtypes = [_P, C.c_char_p, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
lib.dms_lcmlog_rewind.argtypes = [_P]
lib.dms_lcmlog_close.argtypes = [_P]
lib.dms_jpeg_decode.argtypes = [_P, C.c_size_t, C.c_int, C.c_int, _P]
lib.dms_klg_open.argtypes = [C.POINTER(_P), C.c_char_p, C.c_int, C.c_int, C.c_int]
lib.dms_klg_num_frames.argtypes = [_P]
lib.dms_klg_next.argtypes = [_P, _P, _P, C.POINTER(C.c_int64)]
lib.dms_klg_rewind.argtypes = [_P]
lib.dms_klg_close.argtypes = [_P]


class Frame:
    """eflcm.Frame (same attribute names as the generated class, logs/rgbd/eflcm/Frame.py)."""

    def __init__(self, depth=b"", image=b"", timestamp=0, frameNumber=0, senderName="", trackOnly=False, compressed=False, last=False):
        self.trackOnly, self.compressed, self.last = bool(trackOnly), bool(compressed), bool(last)
        self.depth, self.image = bytes(depth), bytes(image)
        self.depthSize, self.imageSize = len(self.depth), len(self.image)
        self.timestamp, self.frameNumber, self.senderName = int(timestamp), int(frameNumber), senderName

    def _msg(self):
        m = FrameMsg()
        m.trackOnly, m.compressed, m.last = int(self.trackOnly), int(self.compressed), int(self.last)
        m.depthSize, m.imageSize = len(self.depth), len(self.image)
        self._keep = (C.create_string_buffer(self.depth, len(self.depth)), C.create_string_buffer(self.image, len(self.image)))
        m.depth = C.cast(self._keep[0], C.c_void_p)
        m.image = C.cast(self._keep[1], C.c_void_p)
        m.timestamp, m.frameNumber = self.timestamp, self.frameNumber
        m.senderName = self.senderName.encode("utf-8")
        return m

    def encode(self):
        m = self._msg()
        n = lib.dms_eflcm_frame_encoded_size(C.byref(m))
        buf = C.create_string_buffer(n)
        w = C.c_size_t(0)
        check(lib.dms_eflcm_frame_encode(C.byref(m), buf, n, C.byref(w)), "dms_eflcm_frame_encode")
        return buf.raw[:w.value]

    @staticmethod
    def decode(data):
        data = bytes(data)
        m = FrameMsg()
        check(lib.dms_eflcm_frame_decode(data, len(data), C.byref(m)), "dms_eflcm_frame_decode")
        f = Frame(C.string_at(m.depth, m.depthSize), C.string_at(m.image, m.imageSize), m.timestamp, m.frameNumber,
                  m.senderName.decode("utf-8", "replace"), m.trackOnly, m.compressed, m.last)
        return f

    def unpack(self, width, height, flipColors=False):
        """(depth u16 HxW, rgb u8 HxWx3) as RawLcmLogReader::getNext produces them."""
        m = self._msg()
        d = np.zeros((height, width), np.uint16)
        rgb = np.zeros((height, width, 3), np.uint8)
        check(lib.dms_frame_unpack(C.byref(m), width, height, int(flipColors), d.ctypes.data_as(_P), rgb.ctypes.data_as(_P)), "dms_frame_unpack")
        return d, rgb


def jpeg_decode(data, width, height):
    """Baseline JPEG -> (H, W, 3) u8 in libjpeg's R, G, B order (dms_jpeg_decode)."""
    data = bytes(data)
    rgb = np.zeros((height, width, 3), np.uint8)
    check(lib.dms_jpeg_decode(data, len(data), width, height, rgb.ctypes.data_as(_P)), "dms_jpeg_decode")
    return rgb


class LcmLogReader:
    """Events of an LCM log (lcm::LogFile): iterate (channel, data bytes, timestamp_us)."""

    def __init__(self, path):
        self.h = _P()
        check(lib.dms_lcmlog_open(C.byref(self.h), path.encode()), "dms_lcmlog_open")

    def __iter__(self):
        ch = C.create_string_buffer(256)
        data, n, ts = _P(), C.c_size_t(0), C.c_int64(0)
        while True:
            rc = lib.dms_lcmlog_next(self.h, ch, 256, C.byref(data), C.byref(n), C.byref(ts))
            if rc == DMS_EOF:
                return
            check(rc, "dms_lcmlog_next")
            yield ch.value.decode(), C.string_at(data, n.value), ts.value

    def rewind(self):
        check(lib.dms_lcmlog_rewind(self.h))

    def close(self):
        if self.h:
            lib.dms_lcmlog_close(self.h)
            self.h = None


class KlgReader:
    """.klg log (RawLogReader): iterate (timestamp, depth u16 HxW, rgb u8 HxWx3)."""

    def __init__(self, path, width, height, flipColors=False):
        self.h = _P()
        self.width, self.height = width, height
        check(lib.dms_klg_open(C.byref(self.h), path.encode(), width, height, int(flipColors)), "dms_klg_open")
        self.numFrames = lib.dms_klg_num_frames(self.h)

    def __iter__(self):
        ts = C.c_int64(0)
        while True:
            d = np.zeros((self.height, self.width), np.uint16)
            rgb = np.zeros((self.height, self.width, 3), np.uint8)
            rc = lib.dms_klg_next(self.h, d.ctypes.data_as(_P), rgb.ctypes.data_as(_P), C.byref(ts))
            if rc == DMS_EOF:
                return
            check(rc, "dms_klg_next")
            yield ts.value, d, rgb

    def rewind(self):
        check(lib.dms_klg_rewind(self.h))

    def close(self):
        if self.h:
            lib.dms_klg_close(self.h)
            self.h = None
