"""Determinism soak of the pipelined session (run on the GPU box): 3 cameras on one GPU, several hundred ticks over a cyclic stream;
the printed hash (pose graphs + final maps) must not depend on the run nor on DMS_SESSION_MAP_STREAMS.
    python scripts/session_soak.py [ticks] [query_from]      (query_from large: three independent maps, no merge)"""
import hashlib, sys, time
sys.path.insert(0, '.')
import numpy as np
from densemonoslam_amd import synth, session, capi
W, H, K = 320, 240, (264.0, 264.0, 160.0, 120.0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = W * H
uniq = 24
def fidx(i):
    p = 2 * (uniq - 1); j = i % p
    return j if j < uniq else p - j
frames = []
for k in range(uniq):
    row = []
    for c, off in enumerate((0, 8, 16)):
        d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
        br, bd = capi.DeviceBuffer(N * 3), capi.DeviceBuffer(N * 2)
        br.upload(np.ascontiguousarray(rgb, np.uint8)); bd.upload(np.ascontiguousarray(d, np.uint16))
        row.append((br, bd))
    frames.append(row)
ns = session.NativeSession(W, H, K, 3, query_from=int(sys.argv[2]) if len(sys.argv) > 2 else 6, model_capacity=6_000_000, fern_photo_thresh=115.0)
st = capi.create_stream()
t0 = time.time()
for k in range(T):
    j = fidx(k)
    ns.step_resident(k, [frames[j][c][0].ptr for c in range(3)], [frames[j][c][1].ptr for c in range(3)], pipelined=True, stream=st)
ns.sync(); capi.lib.dms_stream_sync(st)
h = hashlib.sha256()
pg = ns.pose_graph
for c in sorted(pg):
    for t, p in pg[c]:
        h.update(np.int32(t).tobytes()); h.update(np.asarray(p, np.float32).tobytes())
fo = ns.frame_of
m = np.concatenate([ns.cams[f].model() for f in sorted(set(fo))])
for f in m.dtype.names:
    h.update(np.ascontiguousarray(m[f]).tobytes())
print("ticks", T, "merges", [(a, b, c) for a, b, c, _ in ns.merges], "stats", ns.async_stats(), "surfels", len(m), "sha", h.hexdigest()[:16], "%.1f s" % (time.time() - t0))
