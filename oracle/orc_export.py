"""Test infrastructure: restatement of the reference's end-of-run exports (checker of dms_model_save_ply /
dms_trajectory_save).  PARITY UNPINNED (the reference holds no sample file; ElasticFusion.cpp cannot be built here).

  save_ply_bytes   ElasticFusion::savePly   (ElasticFusion.cpp:781-885) for one map
  trajectory_text  Context::saveTrajectory  (Context.h:117-156)
"""
import struct

import numpy as np

REF_MAX_SENSORS = 3  # Shaders/Vertex.cpp:49


def ref_records(model):
    """orc.SURFEL_DTYPE records -> the reference's 15-float vertex {pos4, col4, times[3], normal + radius}
    (GlobalModel::downloadMap, GlobalModel.cpp:866-896; Shaders/Vertex.cpp:21-50)."""
    out = np.zeros((len(model), 12 + REF_MAX_SENSORS), np.float32)
    out[:, 0:4] = model["pos"]
    out[:, 4:8] = model["col"]
    out[:, 8:8 + REF_MAX_SENSORS] = model["times"][:, :REF_MAX_SENSORS]
    out[:, 8 + REF_MAX_SENSORS:] = model["nrm"]
    return out


def save_ply_bytes(records15, confidenceThreshold, reference_offsets=False):
    rec = np.ascontiguousarray(records15, np.float32)
    count, stride = rec.shape
    flat = rec.reshape(-1)
    valid = int((rec[:, 3] > np.float32(confidenceThreshold)).sum())  # :798-804
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z"
            "\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny\nproperty float nz"
            "\nproperty float radius\nend_header\n" % valid).encode("ascii")  # :806-828
    noff = 18 if reference_offsets else 8 + REF_MAX_SENSORS  # :845-847 reads + 18 (left over from MAX_SENSORS = 10)
    body = bytearray()
    for i in range(count):
        o = i * stride
        if not flat[o + 3] > np.float32(confidenceThreshold):
            continue
        nor = [flat[o + noff + k] if o + noff + k < flat.size else np.float32(0) for k in range(4)]
        c = int(flat[o + 4])  # int(col[0]) (:862-864)
        body += struct.pack("<3f", flat[o], flat[o + 1], flat[o + 2])
        body += bytes([(c >> 16) & 0xFF, (c >> 8) & 0xFF, c & 0xFF])
        body += struct.pack("<4f", np.float32(-1) * nor[0], np.float32(-1) * nor[1], np.float32(-1) * nor[2], nor[3])
    return head + bytes(body)


def trajectory_text(poses):
    """One line per pose: rot / trans of the 3 x 4 matrix row by row, each followed by a blank (Context.h:149-152); an
    std::ostream prints a float with 6 significant digits, like "%g"."""
    lines = []
    for P in poses:
        P = np.asarray(P, np.float32).reshape(4, 4)
        lines.append("".join("%g " % float(P[r, c]) for r in range(3) for c in range(4)) + "\n")
    return "".join(lines)
