import sys, time
sys.path.insert(0, '.')
import numpy as np
from tests.test_session_cpu import SCENARIOS, W, H, K
from tests.test_session_gpu import _make_session
from densemonoslam_amd import synth
from oracle import orc_pipeline, orc
orc.set_threads(16)
sc = SCENARIOS["reference_rule"]
offsets, ticks = (0, 8, 16), int(sys.argv[1]) if len(sys.argv) > 1 else 17
ref = orc_pipeline.Session(3, W, H, K, fern_photo_thresh=sc.fern_photo, wake_latency=3, **sc.opts)
frs = []
for k in range(ticks):
    fr = {}
    for c, off in enumerate(offsets):
        d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
        fr[c] = (rgb, d)
    frs.append(fr)
    ref.step([fr[0], fr[1], fr[2]], k)
s = _make_session("native", sc, 3, capacity=3_000_000)
t0 = time.time()
for k in range(ticks):
    s.step(k, frs[k], pipelined=(k < int(sys.argv[2]) if len(sys.argv) > 2 else True))
print("hip", time.time() - t0)
print("merges", [(m[0], m[1], m[2]) for m in s.merges], [(m[0], m[1], m[2]) for m in ref.merges], s.async_stats(), ref.woken)
pg = s.pose_graph
for c in range(3):
    got, want = pg[c], ref.pose_graph[c]
    bad = [i for i, ((_, a), (_, b)) in enumerate(zip(got, want)) if np.asarray(a, np.float32).tobytes() != np.asarray(b, np.float32).tobytes()]
    print("camera", c, "ticks equal", [t for t, _ in got] == [t for t, _ in want], "poses differing at", bad)
fb = ref.frame_of[0]
m_ref, m_got = ref.cams[fb].model, s.cams[fb].model()
print(len(m_ref), len(m_got))
if len(m_ref) == len(m_got):
    for f in m_ref.dtype.names:
        d = np.nonzero((m_got[f].view(np.uint32) != m_ref[f].view(np.uint32)).reshape(len(m_ref), -1).any(1))[0]
        print(f, len(d), d[:5], d[-5:] if len(d) else "")
