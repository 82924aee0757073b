"""Steady-state cost of a pipelined session tick (dms_session_step_async) for n cameras on this GPU, frames resident, no query due:
    python scripts/session_pipelined_time.py [n_cameras]      (run on the GPU box)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from densemonoslam_amd import synth, session, capi
W, H = 640, 480
K = (528.0, 528.0, 320.0, 240.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = 80
N = W * H
frames = []
for k in range(T):
    row = []
    for c in range(n):
        d, rgb, _ = synth.frame(k + 8 * c, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
        br, bd = capi.DeviceBuffer(N * 3), capi.DeviceBuffer(N * 2)
        br.upload(np.ascontiguousarray(rgb, np.uint8)); bd.upload(np.ascontiguousarray(d, np.uint16))
        row.append((br, bd))
    frames.append(row)
ns = session.NativeSession(W, H, K, n, query_from=10_000, model_capacity=8_000_000)
st = capi.create_stream()
def run(k0, k1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(k0, k1):
        ns.step_resident(k, [frames[k][c][0].ptr for c in range(n)], [frames[k][c][1].ptr for c in range(n)], pipelined=True, stream=st)
    capi.lib.dms_stream_sync(st); torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t) / (k1 - k0)
run(0, 20)
ms = run(20, T); print("cameras", n, "ms per tick %.3f" % ms, "frames/s %.0f" % (n * 1000 / ms))
