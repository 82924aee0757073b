"""three cameras, merges forced at given ticks with the transforms of an oracle session; product contexts beside oracle contexts"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tests.test_session_cpu import SCENARIOS, W, H, K
from densemonoslam_amd import synth, fusion, capi
from oracle import orc_pipeline, orc, orc_ferns
orc.set_threads(16)
sc = SCENARIOS["reference_rule"]
offsets = (0, 8, 16)
M1, M2, END = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frs = []
for k in range(END):
    fr = {}
    for c, off in enumerate(offsets):
        d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
        fr[c] = (rgb, d)
    frs.append(fr)
o = [orc_pipeline.ElasticFusion(W, H, K, timeIdx=c) for c in range(3)]
g = [fusion.ElasticFusion(W, H, K, timeIdx=c, num_sensors=3, model_capacity=3_000_000) for c in range(3)]
sess = orc_pipeline.Session(3, W, H, K, fern_photo_thresh=sc.fern_photo, wake_latency=3, **sc.opts)
for k in range(M2 + 1):
    sess.step([frs[k][0], frs[k][1], frs[k][2]], k)
print("session merges", [(m[0], m[1], m[2]) for m in sess.merges])
TT = {(m[2], m[1]): m[3] for m in sess.merges}
gt = lambda a, b: TT[(a, b)]
gt_unused = lambda a, b: (np.linalg.inv(synth.CORNER_SCENE.pose_fn(offsets[b])) @ synth.CORNER_SCENE.pose_fn(offsets[a])).astype(np.float32)  # map a -> map b

def merge(fb, moving, T):
    owner_o, owner_g = o[fb], g[fb]
    fa = moving[0]
    owner_o.map.model = orc.model_consume(owner_o.map.model, o[fa].map.model, T)
    for c in moving:
        o[c].currPose = orc_ferns._mul44(T, o[c].currPose)
        o[c].map = owner_o.map
        g[c].joinMap(owner_g, T)
        pd = np.empty(16, np.float32)
        g[c].exportPose(pd.ctypes.data) if False else None

def cmp(name, a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    same = a.tobytes() == b.tobytes()
    print("   ", name, "same" if same else "DIFFERENT (%d elements)" % int((a.view(np.uint8) != b.view(np.uint8)).sum()))
    return same

B2B = len(sys.argv) > 4
for k in range(END):
    if B2B:
        ros = [o[c].processFrame(frs[k][c][0], frs[k][c][1]) for c in range(3)]
        for c in range(3):
            ch = g[c].upload_frame(frs[k][c][0], frs[k][c][1])
            g[c].processFrameAsync(g[c]._rgb.ptr, ch, g[c]._depth.ptr)
        for c in range(3):
            rg = g[c].fetch()
            same = np.array(rg.pose, np.float32).tobytes() == ros[c].pose.tobytes()
            if not same or k == END - 1:
                print("tick", k, "camera", c, "pose", "same" if same else "DIFFERENT", "surfels", rg.surfels, ros[c].surfels)
    for c in range(3):
        if B2B:
            break
        if k == END - 1:
            # the view the frame starts from
            o[c].predict(0.7)
            capi.check(fusion.lib.dms_fusion_predict(g[c].h, __import__("ctypes").c_float(0.7), None))
            print("tick", k, "camera", c, "tick counters", o[c].tick, g[c].getOption("tick") if False else "")
            cmp("pred image", g[c].image(9), o[c].pred[0]); cmp("pred vertex", g[c].image(10), o[c].pred[1]); cmp("pred normal", g[c].image(11), o[c].pred[2])
            cmp("fill image", g[c].image(13), o[c].fill[0]); cmp("fill vertex", g[c].image(14), o[c].fill[1])
            mo, mg = o[c].model, g[c].globalModel().downloadMap()
            print("    map before the frame", len(mo), len(mg), all(np.array_equal(mg[f].view(np.uint32), mo[f].view(np.uint32)) for f in mo.dtype.names) if len(mo) == len(mg) else "")
        ro = o[c].processFrame(frs[k][c][0], frs[k][c][1])
        rg = g[c].processFrame(frs[k][c][0], frs[k][c][1])
        same = np.array(rg.pose, np.float32).tobytes() == ro.pose.tobytes()
        if not same or k == END - 1:
            print("tick", k, "camera", c, "pose", "same" if same else "DIFFERENT", "surfels", rg.surfels, ro.surfels)
            if not same:
                print("    pose product", np.array(rg.pose, np.float32).reshape(-1)[:12], "\n    pose oracle ", ro.pose.reshape(-1)[:12])
                print("    track rot/trans same:", np.array(rg.track.rot, np.float32).tobytes() == np.asarray(ro.track.rot, np.float32).tobytes() if hasattr(ro.track, "rot") else "?",
                      "lastA same", np.array(rg.track.lastA).tobytes() == np.array(ro.track.lastA).tobytes(), "so3", rg.track.so3_iterations_run, ro.track.so3_iterations_run,
                      rg.track.lastSO3Error, ro.track.lastSO3Error, rg.track.lastRGBError, ro.track.lastRGBError, "jump", rg.track.rejected_jump, getattr(ro.track, "rejected_jump", None))
            if ro.track is not None:
                print("    oracle track", list(ro.track.iterations_run), ro.track.lastICPError, ro.track.lastICPCount, "product", list(rg.track.iterations_run), rg.track.lastICPError, rg.track.lastICPCount)
    if k in (M1, M2):
        pass
    if k == M1:
        merge(0, [1], gt(1, 0))
    if k == M2:
        merge(2, [0, 1], gt(0, 2))
