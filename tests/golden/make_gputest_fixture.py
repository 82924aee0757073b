"""Builds tests/golden/gputest_pair.npz from the reference's own input fixture.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_gputest_fixture.py
The four PNGs are data files of the reference's GPUTest harness
(elasticfusion/GPUTest/{1c,1d,2c,2d}.png, loaded at GPUTest/src/GPUTest.cpp:30-60); they are
stored here as arrays (depth: raw 16-bit TUM-scale values, *before* the harness' /5 -> mm).
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/elasticfusion/GPUTest"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gputest_pair.npz")

arrs = {}
for n in ("1c", "1d", "2c", "2d"):
    a = np.array(Image.open(os.path.join(SRC, n + ".png")))
    arrs["rgb" + n[0] if n[1] == "c" else "depth" + n[0]] = a.astype(np.uint8 if n[1] == "c" else np.uint16)
np.savez_compressed(OUT, **arrs)
print({k: (v.shape, v.dtype) for k, v in arrs.items()}, os.path.getsize(OUT))
