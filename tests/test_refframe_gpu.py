"""dms_refframe_refine (csrc/refframe.hip) - the second half of ReferenceFrame::resolveRelativeTransformationFern
(ReferenceFrame.h:72-110) - against oracle/orc_pipeline.refine_inter_map on the same two maps: the INACTIVE prediction of the owner's
map at recoveryPose, the refined pose, relativeTransform, the tracker's side outputs and the accept / reject decision at Options'
default thresholds and at thresholds that reject, bit for bit; a second refinement on the same handle (m_rgbd keeps its state: the SO3
pre-alignment then compares with the first call's live image)."""
import numpy as np
import pytest

from tests.test_session_cpu import SCENARIOS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [(320, 240), (640, 480)])  # the session tests' size and BASELINE's
def test_refine_matches_oracle(orc, size):
    from densemonoslam_amd import capi, fusion, synth
    from oracle import orc_pipeline

    assert capi.device_count() >= 1, "no MI355X visible"
    sc = SCENARIOS["reference_rule"]
    W, H = size
    K = (0.825 * W, 0.825 * W, W / 2.0, H / 2.0)
    g = [fusion.ElasticFusion(W, H, K, timeIdx=c, num_sensors=3, model_capacity=2_000_000) for c in range(2)]
    o = [orc_pipeline.ElasticFusion(W, H, K, timeIdx=c) for c in range(2)]
    for k in range(7):
        fr = {}
        for c, off in ((0, 0), (1, sc.offset)):
            d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
            fr[c] = (rgb, d)
        for c in range(2):
            rg = g[c].processFrame(fr[c][0], fr[c][1])
            ro = o[c].processFrame(fr[c][0], fr[c][1])
            assert np.array(rg.pose, np.float32).tobytes() == ro.pose.tobytes()
    # camera 1 queries camera 0's map; the recovery pose a fern match would hand over: camera 1's pose in map 0, a little off
    gt = np.linalg.inv(synth.CORNER_SCENE.pose_fn(0)) @ synth.CORNER_SCENE.pose_fn(6 + sc.offset)
    rec = gt.astype(np.float32)
    rec[0, 3] += np.float32(0.01)
    rf = fusion.ReferenceFrameRefiner(W, H, K)
    rgbd = orc.Odometry(W, H, K[2], K[3], K[0], K[1])
    for call, thr in enumerate([(1e-05, 2e-05, 35000), (1e-05, 2e-05, 10_000_000)]):
        want = orc_pipeline.refine_inter_map(o[0], rgbd, o[1].fill, 1, o[1].tick, o[1].currPose, rec, *thr)
        got = rf.refine(g[0], rec, o[1].currPose, g[1].imagePtr(14), g[1].imagePtr(15), g[1].imagePtr(13), 1, o[1].tick, *thr)
        img, vtx, nrm = rf.prediction()
        assert np.array_equal(img, want["old"][0]) and np.array_equal(vtx.view(np.uint32), want["old"][1].view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), want["old"][2].view(np.uint32))
        assert (vtx[..., 2] > 0).mean() > 0.5  # (every surfel of map 0 is unseen by camera 1: times[1] == -3 passes the INACTIVE cull)
        assert list(got.iterations_run) == list(want["track"].iterations_run) == [50, 50, 50]
        assert got.so3_iterations_run == want["track"].so3_iterations_run
        assert np.float32(got.lastICPError).tobytes() == np.float32(want["lastICPError"]).tobytes()
        assert got.lastICPCount == want["lastICPCount"]
        assert np.array(got.refinedPose, np.float32).tobytes() == want["refinedPose"].tobytes()
        assert np.array(got.relativeTransform, np.float32).tobytes() == want["relativeTransform"].tobytes()
        assert np.allclose([float(v) for v in got.cov_diag], want["cov_diag"], rtol=1e-9, atol=0.0)
        assert bool(got.accepted) == want["accepted"] == (call == 0), (call, got.accepted, want["accepted"])
        assert bool(got.cov_ok) and got.lastICPError < 2e-05 and got.lastICPCount > 35000 * (W * H) / (320 * 240) * 0.9
        # the refinement found the true relative pose again (1 cm off at the start)
        T_gt = gt @ np.linalg.inv(o[1].currPose.astype(np.float64))
        dev = np.abs(np.array(got.relativeTransform, np.float64).reshape(4, 4) - T_gt).max()
        print("size", size, "call", call, "deviation from the ground-truth transform %.2e" % dev)
        assert dev < 1e-2
    rf.close()
    for e in g:
        e.close()
