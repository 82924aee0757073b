"""Degenerate inputs through the whole frame step, HIP (C ABI) against the oracle, bit for bit: frames without a single valid depth
value (in the middle of a run, and as the very first frame: an empty map), depth beyond every cut-off, one lonely valid pixel.  The
tracker then sees no correspondence of either kind (all-zero systems: the pivoted LDL^T's zero-pivot rule decides, identity update),
the index map / fuse / clean run over empty sets, the first fusion after an empty map creates every surfel as new."""
import numpy as np
import pytest

from tests.test_fusion_gpu import fus, surfels_equal, synth  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

W, H = 320, 240
K = (264.0, 264.0, 160.0, 120.0)


def _run(fus, synth, edit, n=7, **opts):
    from oracle import orc_pipeline

    g = fus.ElasticFusion(W, H, K, model_capacity=1_000_000, **opts)
    o = orc_pipeline.ElasticFusion(W, H, K, model_capacity=1_000_000, **opts)
    counts = []
    for k in range(n):
        d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
        d = edit(k, d.copy())
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        assert np.array(rg.pose, np.float32).reshape(4, 4).tobytes() == np.asarray(ro.pose, np.float32).tobytes(), k
        assert int(rg.surfels) == ro.surfels and rg.tick == ro.tick and bool(rg.fused) == ro.fused, (k, rg.surfels, ro.surfels)
        if k > 0:
            assert list(rg.track.iterations_run) == list(ro.track.iterations_run), k
            for f in ("lastICPError", "lastICPCount", "lastRGBError", "lastRGBCount"):
                a, b = np.float32(getattr(rg.track, f)), np.float32(getattr(ro.track, f))
                assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (k, f, a, b)
        counts.append(int(rg.surfels))
    surfels_equal(g.globalModel().downloadMap(), o.model, "final map")
    g.close()
    return counts


def test_blank_depth_frames_in_the_middle_of_a_run(fus, orc, synth):
    def edit(k, d):
        if k in (3, 4):
            d[:] = 0
        return d

    counts = _run(fus, synth, edit)
    assert counts[2] > 10000 and counts[-1] > 10000


def test_first_frame_without_depth_starts_an_empty_map(fus, orc, synth):
    def edit(k, d):
        if k == 0:
            d[:] = 0
        return d

    counts = _run(fus, synth, edit, n=5)
    assert counts[0] == 0 and counts[-1] > 10000


def test_depth_beyond_every_cut_off(fus, orc, synth):
    def edit(k, d):
        if k == 2:
            d[:] = 65535
        if k == 3:
            d[:] = 299  # below the 0.3 m floor of depth_metric.frag / depth_bilateral.frag
        return d

    _run(fus, synth, edit, n=6)


def test_one_valid_pixel(fus, orc, synth):
    def edit(k, d):
        if k == 2:
            v = d[H // 2, W // 2]
            d[:] = 0
            d[H // 2, W // 2] = max(int(v), 1000)
        return d

    _run(fus, synth, edit, n=5)


def test_large_frame_1280x960_takes_the_launch_per_phase_level_0(fus, orc, synth):
    """1 228 800 pixels: more than the 2^19 a resident level-0 grid counts in its pair word, so level 0 runs launch-per-phase while
    levels 1 and 2 stay resident — the same bits as the oracle over three frames."""
    from oracle import orc_pipeline

    W2, H2 = 1280, 960
    K2 = (1056.0, 1056.0, 640.0, 480.0)
    g = fus.ElasticFusion(W2, H2, K2, model_capacity=3_000_000)
    o = orc_pipeline.ElasticFusion(W2, H2, K2, model_capacity=3_000_000)
    for k in range(3):
        d, rgb, _ = synth.frame(k, width=W2, height=H2, K=K2, noise=True)
        rg = g.processFrame(rgb, d)
        ro = o.processFrame(rgb, d)
        assert np.array(rg.pose, np.float32).reshape(4, 4).tobytes() == np.asarray(ro.pose, np.float32).tobytes(), k
        assert int(rg.surfels) == ro.surfels, (k, rg.surfels, ro.surfels)
    surfels_equal(g.globalModel().downloadMap(), o.model, "map at 1280x960")
    g.close()


def test_session_and_helper_entry_points_reject_what_they_cannot_serve():
    """Argument errors of round 5's entry points are reported through the error string, not by a crash: the pipelined step with
    relocalisation on (a lost camera is only seen by a fetch), non-consecutive pipelined ticks, a synchronous step in between (allowed),
    dms_copy_rows_async with a misaligned pitch; dms_copy_rows_async itself moves pitched rows into host-visible memory."""
    import ctypes as C

    from densemonoslam_amd import capi, session, synth

    lib = capi.lib
    lib.dms_copy_rows_async.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
    src, dst = capi.DeviceBuffer(64 * 10), capi.DeviceBuffer(16 * 10)
    a = np.arange(640, dtype=np.uint8)
    src.upload(a)
    assert lib.dms_copy_rows_async(dst.ptr, 16, src.ptr + 8, 64, 16, 10, None) == 0
    got = dst.download(np.uint8, (10, 16))
    assert np.array_equal(got, a.reshape(10, 64)[:, 8:24])
    assert lib.dms_copy_rows_async(dst.ptr, 16, src.ptr + 8, 62, 16, 10, None) != 0 and b"4-byte" in lib.dms_last_error()
    frames = [synth.frame(k, width=W, height=H, K=K, noise=True) for k in range(5)]
    fr = lambda k: {0: (frames[k][1], frames[k][0])}
    s = session.NativeSession(W, H, K, 1, reloc=1)
    with pytest.raises(capi.DmsError, match="relocalisation"):
        s.step(0, fr(0), pipelined=True)
    s.close()
    s = session.NativeSession(W, H, K, 1)
    s.step(0, fr(0), pipelined=True)
    s.step(1, fr(1), pipelined=True)
    with pytest.raises(capi.DmsError, match="consecutive"):
        s.step(3, fr(3), pipelined=True)
    s.step(2, fr(2))  # a synchronous tick completes the pose graph and may follow at any time ...
    s.step(3, fr(3), pipelined=True)  # ... and the pipelined ticks go on behind it
    s.step(4, fr(4), pipelined=True)
    assert [t for t, _ in s.pose_graph[0]] == [1, 2, 3, 4, 5]
    s.close()
