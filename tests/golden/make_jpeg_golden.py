#!/usr/bin/env python3
"""Golden vectors of JPEG colour decoding: streams ENCODED and DECODED by libjpeg itself — Pillow's bundled
libjpeg-turbo, the library family the reference links (GUI/src/Tools/JPEGLoader.h includes jpeglib.h and uses its default
decompression parameters, as Pillow's decoder does outside draft mode): slow integer IDCT, fancy upsampling, fixed-point
colour conversion.  Run in the build container:

    python tests/golden/make_jpeg_golden.py     # writes tests/golden/jpeg_cases.npz

Stored per case: the JPEG bytes and libjpeg's decoded R, G, B scanlines (small cases in full, the 640x480 one as a SHA-256
digest plus its first rows).  csrc/jpeg.hpp must reproduce every byte (tests/test_ingest_cpu.py)."""
import hashlib
import io
import os

import numpy as np
import PIL
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))


def scene(w, h, rng, kind):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    if kind == "smooth":
        img = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 40 + 200 * x / max(w - 1, 1), 255 - 220 * y / max(h - 1, 1)], -1)
    elif kind == "edges":  # saturated colour blocks: sharp chroma edges, values at both ends of the range
        img = np.zeros((h, w, 3))
        img[(x // 5 + y // 3) % 2 == 0] = (255, 0, 0)
        img[(x // 3 + y // 7) % 3 == 1] = (0, 255, 255)
        img[(x + 2 * y) % 11 == 0] = (255, 255, 255)
    else:  # noise
        img = rng.integers(0, 256, (h, w, 3)).astype(np.float64)
    img = img + rng.normal(0, 3.0, img.shape) * (kind == "smooth")
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(20260930)
    cases = [
        # name, w, h, kind, save arguments
        ("q95_420", 64, 48, "smooth", dict(quality=95, subsampling=2)),
        ("q75_420_odd", 50, 37, "edges", dict(quality=75, subsampling=2)),
        ("q90_422", 40, 30, "edges", dict(quality=90, subsampling=1)),
        ("q85_444", 33, 17, "noise", dict(quality=85, subsampling=0)),
        ("q50_420_restart", 64, 48, "noise", dict(quality=50, subsampling=2, restart_marker_blocks=3)),
        ("q100_420_tiny", 17, 9, "noise", dict(quality=100, subsampling=2)),
        ("q30_422_odd_restart_rows", 45, 23, "smooth", dict(quality=30, subsampling=1, restart_marker_rows=1)),
        ("q92_420_optimized", 96, 80, "edges", dict(quality=92, subsampling=2, optimize=True)),
        ("q80_420_2x3", 2, 3, "noise", dict(quality=80, subsampling=2)),    # chroma planes 1, 2 and 3 samples wide: libjpeg
        ("q80_420_4x5", 4, 5, "noise", dict(quality=80, subsampling=2)),    # smooths only planes wider than 2
        ("q80_422_6x4", 6, 4, "noise", dict(quality=80, subsampling=1)),
        ("q95_420_vga", 640, 480, "smooth", dict(quality=95, subsampling=2)),  # OpenCV's default quality, the loggers' size
    ]
    out = {"names": np.array([c[0] for c in cases])}
    for name, w, h, kind, kw in cases:
        img = scene(w, h, rng, kind)
        buf = io.BytesIO()
        Image.fromarray(img, "RGB").save(buf, "JPEG", **kw)
        data = buf.getvalue()
        im = Image.open(io.BytesIO(data))
        assert im.mode == "RGB" and im.size == (w, h)
        dec = np.asarray(im).copy()  # libjpeg's R, G, B scanlines
        out[name + "_jpeg"] = np.frombuffer(data, np.uint8)
        out[name + "_shape"] = np.array([w, h], np.int32)
        out[name + "_sha256"] = np.frombuffer(hashlib.sha256(dec.tobytes()).digest(), np.uint8)
        out[name + "_rgb"] = dec if w * h <= 96 * 80 else dec[:8]
        print(name, len(data), "bytes", dec.shape)
    # a progressive stream: the decoder must refuse it (DMS_ERR_UNSUPPORTED), not mis-decode it
    buf = io.BytesIO()
    Image.fromarray(scene(32, 24, rng, "smooth"), "RGB").save(buf, "JPEG", quality=90, progressive=True)
    out["progressive_jpeg"] = np.frombuffer(buf.getvalue(), np.uint8)
    out["libjpeg"] = np.array(["Pillow %s, libjpeg API %s, libjpeg-turbo %s" % (PIL.__version__, features.version_codec("jpg"),
                                                                                features.version_feature("libjpeg_turbo"))])
    print(out["libjpeg"])
    np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **out)


if __name__ == "__main__":
    main()
