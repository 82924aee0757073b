"""Per-cluster surfel buffers of the frame step (processFrame's `cluster` argument; GlobalModel.h:93-109, GlobalModel.cpp:251-277,
ElasticFusion.cpp:508-515): HIP through the C ABI against oracle/orc_pipeline.py, which restates the reference's rules — a frame that
FUSES under an unknown id starts buffers of its own from the context's feedback buffers (the FIRST frame's unless
computeFeedbackBuffers was called again) and they stay current; a known id switches nothing back.  Bit-exact like every other
frame-step test."""
import numpy as np
import pytest

from tests.test_fusion_gpu import fus, surfels_equal, synth  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

W, H = 320, 240
K = (264.0, 264.0, 160.0, 120.0)


def _run(fus, synth, schedule, refresh_at=()):
    from oracle import orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=1_000_000)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=1_000_000)
    for k, cluster in enumerate(schedule):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        rg = g.processFrame(rgb, d, cluster=cluster)
        ro = o.processFrame(rgb, d, cluster=cluster)
        assert np.array(rg.pose, np.float32).reshape(4, 4).tobytes() == np.asarray(ro.pose, np.float32).tobytes(), k
        assert int(rg.surfels) == ro.surfels, (k, cluster, rg.surfels, ro.surfels)
        ids, cur = g.clusters()
        assert ids == sorted(o.map.clusters) and cur == o.map.current, (k, ids, cur)
        if k in refresh_at:
            g.computeFeedbackBuffers()
            o.computeFeedbackBuffers()
    return g, o


def test_new_cluster_starts_from_the_first_frames_feedback_and_stays_current(fus, orc, synth):
    schedule = [0, 0, 0, 0, 1, 1, 1, 0, 0, 2, 2]  # the 0s after cluster 1 exists switch nothing back
    g, o = _run(fus, synth, schedule)
    assert g.clusters() == ([0, 1, 2], 2)
    for c in (0, 1, 2):
        surfels_equal(g.globalModel(c).downloadMap(), o.map.clusters[c], "cluster %d" % c)
    surfels_equal(g.globalModel().downloadMap(), o.model, "current cluster")
    # the clusters left behind stopped where they were left: 0 after frame 3, 1 after frame 8
    assert len(o.map.clusters[0]) != len(o.map.clusters[1]) != len(o.map.clusters[2])
    g.close()


def test_refreshed_feedback_buffers_seed_the_next_cluster(fus, orc, synth):
    g, o = _run(fus, synth, [0, 0, 0, 0, 0, 7, 7, 7], refresh_at=(3,))
    # cluster 7 was seeded with frame 3's surfels (tick 5 after that frame), not frame 0's
    assert g.clusters() == ([0, 7], 7)
    surfels_equal(g.globalModel().downloadMap(), o.model, "cluster 7")
    assert o.feedback["col"][:, 3].min() == 5.0 and o.feedback["col"][:, 3].max() == 5.0  # the time stamp the buffers were computed at
    g.close()


def test_first_frame_under_another_id_leaves_cluster_0_empty(fus, orc, synth):
    g, o = _run(fus, synth, [3, 3, 3, 0, 0])
    assert g.clusters() == ([0, 3], 3)
    assert g.globalModel(0) is None and len(o.map.clusters[0]) == 0
    surfels_equal(g.globalModel().downloadMap(), o.model, "cluster 3")
    g.close()


def test_new_cluster_on_a_shared_map_is_refused(fus, synth):
    a = fus.ElasticFusion(W, H, K, model_capacity=1_000_000, timeIdx=0)
    b = fus.ElasticFusion(W, H, K, model_capacity=1_000_000, timeIdx=1)
    for k in range(3):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        a.processFrame(rgb, d)
        b.processFrame(rgb, d)
    b.joinMap(a, np.eye(4, dtype=np.float32))
    d, rgb, _ = synth.frame(3, width=W, height=H, K=K, noise=True)
    with pytest.raises(RuntimeError, match="share"):
        b.processFrame(rgb, d, cluster=1)
    b.close()
    a.close()
