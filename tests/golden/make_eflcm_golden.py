#!/usr/bin/env python3
"""Golden vectors of the eflcm.Frame wire format, produced by the REFERENCE's own generated class
(/root/reference/logs/rgbd/eflcm/Frame.py, importable: pure `struct`).  Run in the build
container only (the reference does not exist on the GPU box):

    python tests/golden/make_eflcm_golden.py     # writes tests/golden/eflcm_frames.npz

Stored: for each case the field values and the bytes Frame.encode() produced."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, "/root/reference/logs/rgbd")
from eflcm.Frame import Frame  # noqa: E402  (reference code, imported, never copied)

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260929)
    W, H = 8, 6
    out = {}
    cases = []
    # case 0: raw frame; case 1: compressed flag with zlib depth + opaque "jpeg" bytes, last frame, non-ASCII sender
    depth0 = rng.integers(0, 5000, W * H, dtype=np.uint16)
    rgb0 = rng.integers(0, 256, W * H * 3, dtype=np.uint8)
    cases.append(dict(trackOnly=False, compressed=False, last=False, depth=depth0.tobytes(), image=rgb0.tobytes(), timestamp=1317384506123456789,
                      frameNumber=7, senderName="xtion0"))
    depth1 = (np.arange(W * H, dtype=np.uint16) * 37) % 4000
    cases.append(dict(trackOnly=True, compressed=True, last=True, depth=zlib.compress(depth1.tobytes(), 6), image=bytes(rng.integers(0, 256, 41, dtype=np.uint8)),
                      timestamp=-5, frameNumber=2147483647, senderName="caméra-2"))
    cases.append(dict(trackOnly=False, compressed=False, last=False, depth=b"", image=b"", timestamp=0, frameNumber=0, senderName=""))
    for i, c in enumerate(cases):
        f = Frame()
        f.trackOnly, f.compressed, f.last = c["trackOnly"], c["compressed"], c["last"]
        f.depth, f.image = c["depth"], c["image"]
        f.depthSize, f.imageSize = len(c["depth"]), len(c["image"])
        f.timestamp, f.frameNumber, f.senderName = c["timestamp"], c["frameNumber"], c["senderName"]
        enc = f.encode()
        g = Frame.decode(enc)  # the reference's own round trip
        assert (g.depth, g.image, g.timestamp, g.frameNumber, g.senderName) == (c["depth"], c["image"], c["timestamp"], c["frameNumber"], c["senderName"])
        out["enc%d" % i] = np.frombuffer(enc, np.uint8)
        out["depth%d" % i] = np.frombuffer(c["depth"], np.uint8)
        out["image%d" % i] = np.frombuffer(c["image"], np.uint8)
        out["flags%d" % i] = np.array([c["trackOnly"], c["compressed"], c["last"]], np.int32)
        out["stamp%d" % i] = np.array([c["timestamp"], c["frameNumber"]], np.int64)
        out["sender%d" % i] = np.frombuffer(c["senderName"].encode("utf-8"), np.uint8)
    out["depth1_raw"] = depth1
    out["shape"] = np.array([W, H], np.int32)
    out["fingerprint"] = np.frombuffer(Frame._get_packed_fingerprint(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "eflcm_frames.npz"), **out)
    print("wrote", os.path.join(HERE, "eflcm_frames.npz"), {k: v.shape for k, v in out.items() if k.startswith("enc")})


if __name__ == "__main__":
    main()
