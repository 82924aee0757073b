"""ORACLE (test infrastructure only) — plain-Python restatement of the reference's frame ingest
formats, checked against golden vectors produced by the reference's own generated class
(tests/golden/eflcm_frames.npz, tests/golden/make_eflcm_golden.py): PARITY PINNED for this row.

  eflcm.Frame wire format   logs/rgbd/eflcm/Frame.py:27-65 (fingerprint :66-79)
  LCM event log             liblcm eventlog layout (third party, absent from /root/reference; version unpinned);
                            reader call site GUI/src/Tools/RawLcmLogReader.h:37-85, writer logs/rgbd/iclnuimTolcm.py:32,86
  .klg                      logs/rgbd/RawLogReader.cpp:30 (header), :70-110 (frame)
"""
import struct
import zlib

FINGERPRINT = struct.pack(">Q", ((0x4fb1058a78c4a44b << 1) & 0xffffffffffffffff) + (0x4fb1058a78c4a44b >> 63))


def frame_encode(trackOnly, compressed, last, depth, image, timestamp, frameNumber, senderName):
    name = senderName.encode("utf-8")
    return (FINGERPRINT + struct.pack(">bbbii", trackOnly, compressed, last, len(depth), len(image)) + bytes(depth) + bytes(image) +
            struct.pack(">qi", timestamp, frameNumber) + struct.pack(">I", len(name) + 1) + name + b"\0")


def frame_decode(data):
    if data[:8] != FINGERPRINT:
        raise ValueError("Decode error")
    t, c, l, ds, isz = struct.unpack(">bbbii", data[8:19])
    p = 19
    depth, image = data[p:p + ds], data[p + ds:p + ds + isz]
    p += ds + isz
    ts, fn = struct.unpack(">qi", data[p:p + 12])
    n = struct.unpack(">I", data[p + 12:p + 16])[0]
    name = data[p + 16:p + 16 + n][:-1].decode("utf-8", "replace")
    return dict(trackOnly=bool(t), compressed=bool(c), last=bool(l), depth=depth, image=image, timestamp=ts, frameNumber=fn, senderName=name)


def lcmlog_write(path, events):
    """events: iterable of (timestamp_us, channel, data)"""
    with open(path, "wb") as f:
        for k, (ts, ch, data) in enumerate(events):
            c = ch.encode()
            f.write(struct.pack(">IqqII", 0xEDA1DA01, k, ts, len(c), len(data)) + c + data)


def klg_write(path, frames, compress_depth=False):
    """frames: iterable of (timestamp, depth u16 array, rgb u8 array or None)"""
    frames = list(frames)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for ts, d, rgb in frames:
            db = d.tobytes()
            if compress_depth:
                db = zlib.compress(db, 6)
            ib = b"" if rgb is None else rgb.tobytes()
            f.write(struct.pack("<qii", ts, len(db), len(ib)) + db + ib)
