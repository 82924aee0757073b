"""Golden vectors of the CPU oracle for the fusion half of the frame step (index map, association, fuse,
clean, prediction, fill-in, graph sampling) on a small synthetic stream.

The reference holds no vectors for this path either, so — like oracle_gputest.npz for the tracker —
these pin the ORACLE (tests/test_oracle_cpu.py) and are a second, oracle-free reference for the HIP
path (tests/test_fusion_gpu.py::test_frame_step_reproduces_fusion_golden).  The pose is the
synthetic trajectory's, handed in as the prior with tracking off (hybrid_tracking = 0), so every
stored value is the result of per-element arithmetic only and is reproduced bit for bit.

    python tests/golden/make_fusion_golden.py      (writes tests/golden/oracle_fusion.npz, ~0.3 MB)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

W, H = 160, 120
K = (132.0, 132.0, 80.0, 60.0)
N_FRAMES = 5
KEEP = 2000  # surfels of the final map stored in full


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def frames():
    from densemonoslam_amd import synth

    T0 = None
    for k in range(N_FRAMES):
        d, rgb, T = synth.frame(k, width=W, height=H, K=K, noise=True)
        if T0 is None:
            T0 = T
        yield k, d, rgb, (np.linalg.inv(T0) @ T).astype(np.float32)


def record(out, k, model, index, pred_vertex, pred_image, fill_vertex, fused, fill_in):
    """model: structured surfel array (oracle/orc.py SURFEL_DTYPE layout: pos4 col4 nrm4 times8)."""
    out["count_%d" % k] = np.array([len(model)], np.int64)
    out["map_sha_%d" % k] = sha(np.ascontiguousarray(model).view(np.uint8))
    out["pred_vertex_sha_%d" % k] = sha(pred_vertex)
    out["pred_image_sha_%d" % k] = sha(pred_image)
    out["fill_vertex_sha_%d" % k] = sha(fill_vertex)
    out["flags_%d" % k] = np.array([int(fused), int(fill_in)], np.int64)
    if index is not None:
        out["index_sha_%d" % k] = sha(index)


def compute():
    from oracle import orc, orc_pipeline

    o = orc_pipeline.ElasticFusion(W, H, K, hybrid_tracking=False, model_capacity=200000)
    out = {}
    for k, d, rgb, prior in frames():
        r = o.processFrame(rgb, d, inPose=prior)
        record(out, k, o.model, o.imap[0] if k > 0 else None, o.pred[1], o.pred[0], o.fill[1], r.fused, r.fill_in)
    out["final_index"] = o.imap[0].astype(np.uint32)
    out["final_map_head"] = np.ascontiguousarray(o.model[:KEEP]).view(np.uint8).reshape(KEEP, -1).copy()
    out["graph_samples_97"] = orc.sample_graph(o.model, 97)
    return out


if __name__ == "__main__":
    res = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_fusion.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes;", int(res["count_%d" % (N_FRAMES - 1)][0]), "surfels")
