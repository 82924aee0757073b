"""The collaborative session past the merge on the MI355X (densemonoslam_amd/session.py with the product behind it), against the
one-process oracle session (oracle/orc_pipeline.Session): publish -> match -> verify -> merge -> 20 more frames; merged map, both
trajectories and the merge transform must be the same bits.
  * one process, two cameras on one device: the same-rank path (dms_fusion_join_map, dms_ferns_consume) - what the reference does;
  * two ranks over gloo sharing the one GPU of the test box: the cross-rank path (dms_model_export_records / _consume_records,
    dms_ferns_export_records / _consume_records, dms_fusion_import_camera, frame forwarding).  (RCCL needs one GPU per rank; with
    it the transport tensors live in HBM and nothing else changes: session.CollabSession(device=cuda).)"""
import multiprocessing as mp
import os

import numpy as np
import pytest

from tests.test_session_cpu import (FOUR_OFFSETS, FOUR_TICKS, FOUR_TICKS_PIPELINED, SCENARIOS, H, K, W, _free_port, check_four, frames_at, run_oracle_session,
                                    run_oracle_session_n, spawn)

pytestmark = pytest.mark.gpu
# "reference_rule": the merge is decided by Ferns::findFrame(interMap = 1) + the full-resolution refinement and acceptance of
# ReferenceFrame.h:72-110 at Options' default thresholds (dms_refframe_refine on the owner's GPU); "thumbnail_only": rounds 3-4's
# inter_map = 2 session without the refinement, kept as an extra (tests/test_session_cpu.py)
_ORACLE = {}


def n_ticks(scenario):
    return SCENARIOS[scenario].query_from + 21


@pytest.fixture(params=["reference_rule", "thumbnail_only"])
def oracle_session(orc, request):
    name = request.param
    if name not in _ORACLE:
        s = run_oracle_session(name, n_ticks(name), relative_cons=True)
        assert len(s.merges) == 1 and n_ticks(name) - 1 - s.merges[0][0] >= 20, s.merges
        if SCENARIOS[name].opts.get("full_refine"):
            assert [r[3] for r in s.refinements] == [True], s.refinements
        _ORACLE[name] = s
    s = _ORACLE[name]
    s.scenario = name
    return s


def _check(ref, host, merges, N_TICKS=None):
    N_TICKS = N_TICKS or n_ticks(ref.scenario)
    k_merge, fb, fa, T = ref.merges[0]
    assert [(m[0], m[1], m[2]) for m in merges] == [(k_merge, fb, fa)], merges
    assert np.asarray(merges[0][3], np.float32).tobytes() == T.tobytes(), "the relative transform differs from the oracle's"
    m_ref, m_got = ref.cams[fb].model, host["map"]
    assert len(m_got) == len(m_ref), (len(m_got), len(m_ref))
    for f in m_ref.dtype.names:
        assert np.array_equal(m_got[f].view(np.uint32), m_ref[f].view(np.uint32)), "merged map differs in field " + f
    assert host["fern_frames"] == len(ref.ferns[fb].frames)
    for c in range(2):
        got, want = host["pose_graph"][c], ref.pose_graph[c]
        assert [t for t, _ in got] == [t for t, _ in want] and len(got) == N_TICKS
        for i, ((_, a), (_, b)) in enumerate(zip(got, want)):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes(), "camera %d pose %d differs" % (c, i)
        # Context::relativeCons(): the consumed camera's row re-based by the merge (ReferenceFrame.h:133-136), the other one untouched
        rc = host["relative_cons"][c]
        assert len(rc) == 1 and np.asarray(rc[0], np.float32).tobytes() == ref.relative_cons[c][0].tobytes(), "camera %d: relative constraint differs" % c


def _make_session(impl, sc, n, rank=0, world=1, capacity=2_000_000):
    """impl "python": densemonoslam_amd.session.CollabSession (the protocol in Python over torch.distributed, product engines behind it);
    "native": dms_session (include/dmslam_session.h: the same protocol compiled into the library, what a C++ front end links)."""
    import torch

    from densemonoslam_amd import session

    row = lambda c: np.arange(6, dtype=np.float32) * np.float32(0.25 + c)  # a constraint row the caller's solver produced before any merge
    if impl == "native":
        tr = session.TorchTransport(rank, world) if world > 1 else None
        s = session.NativeSession(W, H, K, n, rank=rank, world=world, transport=tr, fern_photo_thresh=sc.fern_photo, model_capacity=capacity,
                                  **sc.opts)
        for c in s.hosted():
            s.addRelativeConstraint(c, row(c)[:3], row(c)[3:])
        return s
    be = session.GpuBackend(W, H, K, torch.device("cuda", 0), fern_opts=dict(photoThresh=sc.fern_photo), model_capacity=capacity)
    s = session.CollabSession(be, n, W, H, rank=rank, world=world, **sc.opts)
    for c in s.hosted():
        s.relative_cons[c].append(row(c))
    return s


@pytest.mark.parametrize("impl", ["native", "python"])
def test_two_cameras_one_device_merge_and_continue(oracle_session, impl):
    from densemonoslam_amd import capi, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    sc = SCENARIOS[oracle_session.scenario]
    s = _make_session(impl, sc, 2)
    for k in range(n_ticks(sc.name)):
        s.step(k, sc.frames(synth, k))
    assert s.refinements == [r[:4] for r in oracle_session.refinements]
    fb = oracle_session.merges[0][1]
    host = dict(map=s.cams[fb].model(), fern_frames=len(s.ferns[fb]), pose_graph=s.pose_graph, relative_cons=s.relative_cons)
    _check(oracle_session, host, s.merges)
    s.close()


def _worker(rank, world, port, q, scenario, impl, pipelined=False, ticks=None):
    sc = SCENARIOS[scenario]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["DMS_TRACK_MODE"] = "launches"  # two processes on one device must not spin side by side
    import torch
    import torch.distributed as dist

    from densemonoslam_amd import session, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = _make_session(impl, sc, 2, rank, world)
    for k in range(ticks or n_ticks(scenario)):
        fr = sc.frames(synth, k)
        s.step(k, {c: fr[c] for c in fr if c % world == rank}, **({"pipelined": True} if pipelined else {}))
    res = dict(rank=rank, merges=s.merges, hosted=s.hosted(), refinements=s.refinements, stats=s.async_stats() if pipelined else None)
    if s.hosted():
        fb = s.frame_of[s.hosted()[0]]
        res.update(map=s.cams[fb].model(), fern_frames=len(s.ferns[fb]), pose_graph={c: s.pose_graph[c] for c in s.hosted()},
                   relative_cons={c: s.relative_cons[c] for c in s.hosted()})
    q.put(res)
    dist.barrier()
    s.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("impl", ["native", "python"])
def test_two_ranks_merge_across_processes_and_continue(oracle_session, impl):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, oracle_session.scenario, impl)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r["rank"]: r for r in [q.get(timeout=900) for _ in range(world)]}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    fb = oracle_session.merges[0][1]
    hb = fb % world
    assert results[hb]["hosted"] == [0, 1] and results[1 - hb]["hosted"] == []
    assert [(m[0], m[1], m[2]) for m in results[1 - hb]["merges"]] == [(m[0], m[1], m[2]) for m in results[hb]["merges"]]
    for r in range(world):  # the cross-rank refinement (fill-in textures point to point, result broadcast): the same list on every rank
        assert results[r]["refinements"] == [x[:4] for x in oracle_session.refinements]
    _check(oracle_session, results[hb], results[hb]["merges"])


@pytest.mark.parametrize("impl", ["native", "python"])
def test_three_cameras_chained_merge_one_device(orc, impl):
    """A reference frame that already holds two cameras is consumed by a third map (ReferenceFrame::consumeReferenceFrame moves EVERY
    camera of the consumed frame, ReferenceFrame.h:127-145): frame 0 consumes camera 1's map at tick 6, then frame 2 consumes frame 0 -
    its founder carries the surfels over, the camera that had joined it only moves (dms_fusion_join_map on an already joined
    context).  Same bits as the one-process oracle session 5 frames past the second merge."""
    import torch

    from densemonoslam_amd import session, synth
    from oracle import orc_pipeline

    sc = SCENARIOS["reference_rule"]
    offsets, ticks = (0, 8, 16), 13

    def frames(k):
        out = {}
        for c, off in enumerate(offsets):
            d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
            out[c] = (rgb, d)
        return out

    ref = orc_pipeline.Session(3, W, H, K, fern_photo_thresh=sc.fern_photo, **sc.opts)
    s = _make_session(impl, sc, 3, capacity=3_000_000)
    for k in range(ticks):
        fr = frames(k)
        ref.step([fr[0], fr[1], fr[2]], k)
        s.step(k, fr)
    assert len(ref.merges) == 2 and ref.frame_of == [ref.merges[1][1]] * 3, (ref.merges, ref.frame_of)
    k2, fb2, fa2, _ = ref.merges[1]
    assert ticks - 1 - k2 >= 5
    # the second merge consumed a frame that held two cameras
    assert ref.merges[0][1] == fa2, ref.merges
    assert [(m[0], m[1], m[2]) for m in s.merges] == [(m[0], m[1], m[2]) for m in ref.merges]
    for got, want in zip(s.merges, ref.merges):
        assert np.asarray(got[3], np.float32).tobytes() == want[3].tobytes()
    assert s.frame_of == ref.frame_of and s.refinements == [r[:4] for r in ref.refinements]
    m_ref, m_got = ref.cams[fb2].model, s.cams[fb2].model()
    assert len(m_got) == len(m_ref), (len(m_got), len(m_ref))
    for f in m_ref.dtype.names:
        assert np.array_equal(m_got[f].view(np.uint32), m_ref[f].view(np.uint32)), "merged map differs in field " + f
    for c in range(3):
        got, want = s.pose_graph[c], ref.pose_graph[c]
        assert [t for t, _ in got] == [t for t, _ in want] and len(got) == ticks
        for i, ((_, a), (_, b)) in enumerate(zip(got, want)):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes(), "camera %d pose %d differs" % (c, i)
    s.close()


def test_native_session_over_a_one_rank_rccl_communicator(orc):
    """The compiled session with the RCCL transport (dms_transport_rccl): a one-rank communicator is what a one-GPU box can form - both
    cameras live on rank 0, and every collective of the protocol (block all-gather, table all-gather, the refinement's broadcast) is a
    real ncclAllGather.  Same merge, same transform bits as the one-process oracle session."""
    from densemonoslam_amd import capi, collab, session, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    sc = SCENARIOS["reference_rule"]
    ref = _ORACLE.get("reference_rule") or run_oracle_session("reference_rule", n_ticks("reference_rule"), relative_cons=False)
    tr = session.RcclTransport(collab.RcclCarrier(0, 1, collab.RcclCarrier.unique_id()))
    s = session.NativeSession(W, H, K, 2, rank=0, world=1, transport=tr, fern_photo_thresh=sc.fern_photo, model_capacity=2_000_000, **sc.opts)
    for k in range(ref.merges[0][0] + 3):
        s.step(k, sc.frames(synth, k))
    assert [(m[0], m[1], m[2]) for m in s.merges] == [(m[0], m[1], m[2]) for m in ref.merges]
    assert s.merges[0][3].tobytes() == ref.merges[0][3].tobytes() and s.refinements == [r[:4] for r in ref.refinements]
    s.close()


def _worker3(rank, world, port, q, impl, ticks, offsets, pipelined=False):
    sc = SCENARIOS["reference_rule"]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["DMS_TRACK_MODE"] = "launches"  # two processes on one device must not spin side by side
    import torch.distributed as dist

    from densemonoslam_amd import synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = _make_session(impl, sc, len(offsets), rank, world, capacity=3_000_000)
    for k in range(ticks):
        fr = {}
        for c, off in enumerate(offsets):
            if c % world == rank:
                d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
                fr[c] = (rgb, d)
        s.step(k, fr, **({"pipelined": True} if pipelined else {}))
    res = dict(rank=rank, merges=s.merges, hosted=s.hosted(), refinements=s.refinements, frame_of=s.frame_of)
    if s.hosted():
        fb = s.frame_of[s.hosted()[0]]
        res.update(map=s.cams[fb].model(), pose_graph={c: s.pose_graph[c] for c in s.hosted()})
    q.put(res)
    dist.barrier()
    s.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("impl", ["native", "python"])
def test_three_cameras_two_ranks_chained_merge(orc, impl):
    """Cameras 0 and 2 are read on rank 0, camera 1 on rank 1 (two slots per rank in the block all-gather).  Tick 6: frame 0 (rank 0)
    consumes camera 1's map ACROSS ranks (records, import, frame forwarding from then on); tick 7: frame 2 consumes frame 0, which
    now holds its founder and an IMPORTED camera, on the same rank (dms_fusion_join_map of an imported context).  Map, the three
    trajectories and both transforms are the one-process oracle session's bit for bit."""
    from densemonoslam_amd import synth
    from oracle import orc_pipeline

    sc = SCENARIOS["reference_rule"]
    offsets, ticks, world = (0, 8, 16), 12, 2
    ref = orc_pipeline.Session(3, W, H, K, fern_photo_thresh=sc.fern_photo, **sc.opts)
    for k in range(ticks):
        ref.step([tuple(reversed(synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)[:2])) for off in offsets], k)
    assert [(m[1], m[2]) for m in ref.merges] == [(0, 1), (2, 0)], ref.merges
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, world, port, q, impl, ticks, offsets)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r["rank"]: r for r in [q.get(timeout=900) for _ in range(world)]}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert [(m[0], m[1], m[2]) for m in results[r]["merges"]] == [(m[0], m[1], m[2]) for m in ref.merges]
        for got, want in zip(results[r]["merges"], ref.merges):
            assert np.asarray(got[3], np.float32).tobytes() == want[3].tobytes()
        assert results[r]["frame_of"] == ref.frame_of and results[r]["refinements"] == [x[:4] for x in ref.refinements]
    assert results[0]["hosted"] == [0, 1, 2] and results[1]["hosted"] == []
    m_ref, m_got = ref.cams[2].model, results[0]["map"]
    assert len(m_got) == len(m_ref), (len(m_got), len(m_ref))
    for f in m_ref.dtype.names:
        assert np.array_equal(m_got[f].view(np.uint32), m_ref[f].view(np.uint32)), "merged map differs in field " + f
    for c in range(3):
        got, want = results[0]["pose_graph"][c], ref.pose_graph[c]
        assert [t for t, _ in got] == [t for t, _ in want] and len(got) == ticks
        for i, ((_, a), (_, b)) in enumerate(zip(got, want)):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes(), "camera %d pose %d differs" % (c, i)


# ---- the pipelined step (dms_session_step_async): no host synchronisation, the full query only three ticks after a descriptor hit ----
PIPE_TICKS = 21


@pytest.fixture
def pipelined_oracle(orc):
    if "pipelined" not in _ORACLE:
        s = run_oracle_session("reference_rule", PIPE_TICKS, relative_cons=True, wake_latency=3)
        q = SCENARIOS["reference_rule"].query_from
        # the descriptor search hits at the first tick that may query; the reference's block runs three ticks later, verifies and merges
        assert s.hits[q] and not any(s.hits[k] for k in range(q)) and s.woken == [q + 3], (s.hits, s.woken)
        assert [(m[0], m[1], m[2]) for m in s.merges] == [(q + 3, 0, 1)] and [r[3] for r in s.refinements] == [True], (s.merges, s.refinements)
        s.scenario = "reference_rule"
        _ORACLE["pipelined"] = s
    return _ORACLE["pipelined"]


def test_pipelined_session_one_device(pipelined_oracle):
    """Every tick through dms_session_step_async: the frames pipeline, nothing is fetched, pose graphs arrive from the gathered blocks
    two ticks late; one tick wakes (the search of tick 6 hit), runs the reference's block, merges - then both cameras go on in one map
    without a host round trip.  The same bits as the oracle session on the same schedule (Session(wake_latency = 3))."""
    from densemonoslam_amd import capi, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    ref = pipelined_oracle
    sc = SCENARIOS["reference_rule"]
    s = _make_session("native", sc, 2)
    for k in range(PIPE_TICKS):
        s.step(k, sc.frames(synth, k), pipelined=True)
    assert s.async_stats() == {"ticks": PIPE_TICKS, "woken": len(ref.woken)}
    assert s.refinements == [r[:4] for r in ref.refinements]
    fb = ref.merges[0][1]
    host = dict(map=s.cams[fb].model(), fern_frames=len(s.ferns[fb]), pose_graph=s.pose_graph, relative_cons=s.relative_cons)
    _check(ref, host, s.merges, PIPE_TICKS)
    s.close()


def test_pipelined_and_synchronous_steps_mix(pipelined_oracle):
    """Ticks 0 - 4 pipelined, 5 - 8 synchronous: the synchronous step completes the pose graphs first and queries every pair itself, so the
    session merges where the all-synchronous one does (tick 6), and the pipelined ticks behind it wake nobody."""
    from densemonoslam_amd import synth

    sc = SCENARIOS["reference_rule"]
    ref = _ORACLE.get("reference_rule") or run_oracle_session("reference_rule", n_ticks("reference_rule"), relative_cons=False)
    s = _make_session("native", sc, 2)
    for k in range(12):
        s.step(k, sc.frames(synth, k), pipelined=not (5 <= k <= 8))
    assert [(m[0], m[1], m[2]) for m in s.merges] == [(m[0], m[1], m[2]) for m in ref.merges]
    assert s.merges[0][3].tobytes() == ref.merges[0][3].tobytes() and s.async_stats() == {"ticks": 8, "woken": 0}
    for c in range(2):
        got, want = s.pose_graph[c], ref.pose_graph[c][:12]
        assert [t for t, _ in got] == [t for t, _ in want]
        for (_, a), (_, b) in zip(got, want):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes()
    s.close()


def test_pipelined_session_two_ranks(pipelined_oracle):
    """The same across two ranks (gloo transport, one GPU): the hit rows ride the all-gather, both ranks wake at the same tick, the map
    crosses ranks, and from then on rank 1 only forwards its camera's frames - without a synchronisation of its own."""
    ref = pipelined_oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "reference_rule", "native", True, PIPE_TICKS)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r["rank"]: r for r in [q.get(timeout=900) for _ in range(world)]}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    fb = ref.merges[0][1]
    hb = fb % world
    assert results[hb]["hosted"] == [0, 1] and results[1 - hb]["hosted"] == []
    for r in range(world):
        assert results[r]["stats"] == {"ticks": PIPE_TICKS, "woken": len(ref.woken)}
        assert [(m[0], m[1], m[2]) for m in results[r]["merges"]] == [(m[0], m[1], m[2]) for m in ref.merges]
        assert results[r]["refinements"] == [x[:4] for x in ref.refinements]
    _check(ref, results[hb], results[hb]["merges"], PIPE_TICKS)


def test_pipelined_three_cameras_two_ranks(orc):
    """Three cameras on two ranks (two slots per rank, an empty one on rank 1; two databases on rank 0), chained merges, pipelined:
    same bits as the oracle session on the same schedule."""
    from densemonoslam_amd import synth
    from oracle import orc_pipeline

    sc = SCENARIOS["reference_rule"]
    offsets, ticks, world = (0, 8, 16), 17, 2
    ref = orc_pipeline.Session(3, W, H, K, fern_photo_thresh=sc.fern_photo, wake_latency=3, **sc.opts)
    for k in range(ticks):
        ref.step([tuple(reversed(synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)[:2])) for off in offsets], k)
    print("woken", ref.woken, "merges", [(m[0], m[1], m[2]) for m in ref.merges])
    assert len(ref.merges) == 2 and len(set(ref.frame_of)) == 1, ref.merges
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, world, port, q, "native", ticks, offsets, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r["rank"]: r for r in [q.get(timeout=900) for _ in range(world)]}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert [(m[0], m[1], m[2]) for m in results[r]["merges"]] == [(m[0], m[1], m[2]) for m in ref.merges]
        for got, want in zip(results[r]["merges"], ref.merges):
            assert np.asarray(got[3], np.float32).tobytes() == want[3].tobytes()
        assert results[r]["frame_of"] == ref.frame_of and results[r]["refinements"] == [x[:4] for x in ref.refinements]
    host = [r for r in range(world) if results[r]["hosted"]]
    assert len(host) == 1 and results[host[0]]["hosted"] == [0, 1, 2]
    fb = ref.frame_of[0]
    m_ref, m_got = ref.cams[fb].model, results[host[0]]["map"]
    assert len(m_got) == len(m_ref), (len(m_got), len(m_ref))
    for f in m_ref.dtype.names:
        assert np.array_equal(m_got[f].view(np.uint32), m_ref[f].view(np.uint32)), "merged map differs in field " + f
    for c in range(3):
        got, want = results[host[0]]["pose_graph"][c], ref.pose_graph[c]
        assert [t for t, _ in got] == [t for t, _ in want] and len(got) == ticks
        for i, ((_, a), (_, b)) in enumerate(zip(got, want)):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes(), "camera %d pose %d differs" % (c, i)


def test_a_merged_camera_keeps_its_pose_bottom_row(orc):
    """relativeTransform = recoveryPose * currPose.inverse() is a general 4 x 4 inverse (ReferenceFrame.h:95): its bottom row need not be
    exactly (0 0 0 1), currPose = relativeTransform * currPose (:131) inherits it, and the tracker only ever assigns the top three rows
    (ElasticFusion.cpp:246-247) - so it stays with the camera and every pose.inverse() sees it.  (Found by the pipelined schedule: a
    merge at tick 13 whose transform had such a row.)  Two cameras, a join with a perturbed bottom row, three more frames: same bits."""
    from densemonoslam_amd import fusion, synth
    from oracle import orc_ferns, orc_pipeline

    sc = SCENARIOS["reference_rule"]
    g = [fusion.ElasticFusion(W, H, K, timeIdx=c, num_sensors=3, model_capacity=2_000_000) for c in range(2)]
    o = [orc_pipeline.ElasticFusion(W, H, K, timeIdx=c) for c in range(2)]
    T = (np.linalg.inv(synth.CORNER_SCENE.pose_fn(0)) @ synth.CORNER_SCENE.pose_fn(sc.offset)).astype(np.float32)  # map 1 -> map 0
    T[3] = np.array([3e-9, -2e-9, 1e-9, 0.99999994], np.float32)
    for k in range(7):
        for c, off in ((0, 0), (1, sc.offset)):
            d, rgb, _ = synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)
            ro, rg = o[c].processFrame(rgb, d), g[c].processFrame(rgb, d)
            assert np.array(rg.pose, np.float32).tobytes() == ro.pose.tobytes(), (k, c)
            if k > 3 and c == 1:
                assert ro.pose[3].tobytes() != np.array([0, 0, 0, 1], np.float32).tobytes()  # (the row really is carried)
        if k == 3:
            o[0].map.model = orc.model_consume(o[0].map.model, o[1].map.model, T)
            o[1].currPose = orc_ferns._mul44(T, o[1].currPose)
            o[1].map = o[0].map
            g[1].joinMap(g[0], T)
    m_ref, m_got = o[0].model, g[0].globalModel().downloadMap()
    assert len(m_ref) == len(m_got)
    for f in m_ref.dtype.names:
        assert np.array_equal(m_got[f].view(np.uint32), m_ref[f].view(np.uint32)), "map differs in field " + f
    g[1].close()
    g[0].close()


def test_pipelined_wakes_continue_while_refinements_reject(orc):
    """Acceptance thresholds nothing passes (icpCountThresh above the pixel count): every search from query_from on hits, so every tick
    from query_from + 3 on is woken - the hit rows that arrive DURING a woken tick count like any other - each runs the reference's
    block, refines, rejects; no merge.  A dms_session_sync between two ticks (pose graphs read mid-run) must not lose a hit either."""
    from densemonoslam_amd import synth

    sc = SCENARIOS["reference_rule"]
    ticks = sc.query_from + 8
    opts = dict(sc.opts, icp_count_thresh=10_000_000)
    from oracle import orc_pipeline

    ref = orc_pipeline.Session(2, W, H, K, fern_photo_thresh=sc.fern_photo, wake_latency=3, **opts)
    for k in range(ticks):
        fr = sc.frames(synth, k)
        ref.step([fr[0], fr[1]], k)
    assert ref.woken == list(range(sc.query_from + 3, ticks)) and not ref.merges and len(ref.refinements) >= 2, (ref.woken, [r[:4] for r in ref.refinements])
    import copy

    sc2 = copy.copy(sc)
    sc2.opts = opts
    s = _make_session("native", sc2, 2)
    for k in range(ticks):
        s.step(k, sc.frames(synth, k), pipelined=True)
        if k == sc.query_from + 1:
            assert len(s.pose_graph[0]) == k + 1  # (dms_session_pose_graph drains the outstanding entries)
    assert s.async_stats() == {"ticks": ticks, "woken": len(ref.woken)} and s.merges == []
    assert s.refinements == [r[:4] for r in ref.refinements]
    for c in range(2):
        got, want = s.pose_graph[c], ref.pose_graph[c]
        assert [t for t, _ in got] == [t for t, _ in want] and len(got) == ticks
        for (_, a), (_, b) in zip(got, want):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes()
    s.close()


def test_python_session_model_on_the_pipelined_schedule(pipelined_oracle):
    """densemonoslam_amd.session.CollabSession(wake_latency = 3) - the model the CPU tests run over gloo - with the product's engines
    behind it (its searchHit = dms_ferns_search_blocks_hd on one block): the same wake, merge and bits as the oracle session."""
    import torch

    from densemonoslam_amd import session, synth

    ref = pipelined_oracle
    sc = SCENARIOS["reference_rule"]
    be = session.GpuBackend(W, H, K, torch.device("cuda", 0), fern_opts=dict(photoThresh=sc.fern_photo), model_capacity=2_000_000)
    s = session.CollabSession(be, 2, W, H, wake_latency=3, **sc.opts)
    for k in range(ref.merges[0][0] + 3):
        s.step(k, sc.frames(synth, k))
    assert s.woken == ref.woken and [(m[0], m[1], m[2]) for m in s.merges] == [(m[0], m[1], m[2]) for m in ref.merges]
    assert np.asarray(s.merges[0][3], np.float32).tobytes() == ref.merges[0][3].tobytes()
    s.close()


@pytest.mark.parametrize("size", ["640x480", "1241x376"])
def test_pipelined_session_at_the_baseline_sizes(orc, size):
    """BASELINE's frame sizes (TUM 640 x 480; KITTI 1241 x 376 with its intrinsics: 155 x 47 thumbnails, blocks whose size is no multiple
    of 16 bytes): two cameras, the pipelined tick, against the oracle session on the same schedule - the wake at tick 9, the merge (fern
    match under interMap = 1, full-resolution refinement accepted at Options' thresholds), the transform, both trajectories and the
    merged map three frames past it, bit for bit."""
    from densemonoslam_amd import session, synth
    from oracle import orc_pipeline

    sc = SCENARIOS["reference_rule"]
    W2, H2, K2 = (640, 480, synth.K_640) if size == "640x480" else (1241, 376, synth.K_KITTI)
    ticks = sc.query_from + 7
    frames = []
    for k in range(ticks):
        fr = {}
        for c, off in ((0, 0), (1, sc.offset)):
            d, rgb, _ = synth.frame(k + off, width=W2, height=H2, K=K2, noise=True, scene=synth.CORNER_SCENE)
            fr[c] = (rgb, d)
        frames.append(fr)
    ref = orc_pipeline.Session(2, W2, H2, K2, fern_photo_thresh=sc.fern_photo, wake_latency=3, **sc.opts)
    for k in range(ticks):
        ref.step([frames[k][0], frames[k][1]], k)
    assert ref.woken == [sc.query_from + 3] and [(m[0], m[1], m[2]) for m in ref.merges] == [(sc.query_from + 3, 1, 0)], (ref.woken, ref.merges)
    assert [r[3] for r in ref.refinements] == [True]
    s = session.NativeSession(W2, H2, K2, 2, fern_photo_thresh=sc.fern_photo, model_capacity=4_000_000, **sc.opts)
    for k in range(ticks):
        s.step(k, frames[k], pipelined=True)
    assert s.async_stats() == {"ticks": ticks, "woken": 1} and s.refinements == [r[:4] for r in ref.refinements]
    assert [(m[0], m[1], m[2]) for m in s.merges] == [(m[0], m[1], m[2]) for m in ref.merges]
    assert s.merges[0][3].tobytes() == ref.merges[0][3].tobytes()
    fb = ref.merges[0][1]
    m_ref, m_got = ref.cams[fb].model, s.cams[fb].model()
    assert len(m_got) == len(m_ref) > 1_000_000, (len(m_got), len(m_ref))
    for f in m_ref.dtype.names:
        assert np.array_equal(m_got[f].view(np.uint32), m_ref[f].view(np.uint32)), "merged map differs in field " + f
    for c in range(2):
        got, want = s.pose_graph[c], ref.pose_graph[c]
        assert [t for t, _ in got] == [t for t, _ in want] and len(got) == ticks
        for i, ((_, a), (_, b)) in enumerate(zip(got, want)):
            assert np.asarray(a, np.float32).tobytes() == np.asarray(b, np.float32).tobytes(), "camera %d pose %d differs" % (c, i)
    s.close()


def test_pipelined_result_does_not_depend_on_the_stream_arrangement(monkeypatch):
    """Three cameras on one device, 150 pipelined ticks over a cyclic stream, two merges on the way: pose graphs and the final map are the
    same bytes whether everything runs on the caller's stream or the maps take a pool of two or three streams (a race between a map's
    stream and the exchange would show here long before it shows against the oracle's 20 ticks)."""
    import hashlib

    from densemonoslam_amd import session, synth

    sc = SCENARIOS["reference_rule"]
    uniq, ticks = 24, 150
    frames = [{c: tuple(reversed(synth.frame(k + off, width=W, height=H, K=K, noise=True, scene=synth.CORNER_SCENE)[:2])) for c, off in enumerate((0, 8, 16))}
              for k in range(uniq)]

    def run(pool):
        monkeypatch.setenv("DMS_SESSION_MAP_STREAMS", str(pool))
        s = session.NativeSession(W, H, K, 3, fern_photo_thresh=sc.fern_photo, model_capacity=3_000_000, **sc.opts)
        for k in range(ticks):
            j = k % (2 * (uniq - 1))
            s.step(k, frames[j if j < uniq else 2 * (uniq - 1) - j], pipelined=True)
        h = hashlib.sha256()
        for c, rows in sorted(s.pose_graph.items()):
            for t, p in rows:
                h.update(np.int32(t).tobytes() + np.asarray(p, np.float32).tobytes())
        m = s.cams[s.frame_of[0]].model()
        for f in m.dtype.names:
            h.update(np.ascontiguousarray(m[f]).tobytes())
        out = (h.hexdigest(), [(a, b, c) for a, b, c, _ in s.merges], s.async_stats(), len(m))
        s.close()
        return out

    ref = run(0)
    assert len(ref[1]) == 2 and ref[2]["ticks"] == ticks, ref
    for pool in (2, 3):
        assert run(pool) == ref, pool


# ---- BASELINE config 5 at its stated size: --n 4, four cameras, all fusing (MainController.cpp:229,262-400) -------------------------
_FOUR = {}


def four_oracle(wake=None, **opts):
    key = (wake, tuple(sorted(opts.items())))
    if key not in _FOUR:
        ticks = FOUR_TICKS if wake is None else FOUR_TICKS_PIPELINED
        _FOUR[key] = (run_oracle_session_n(FOUR_OFFSETS, ticks, **({"wake_latency": wake} if wake else {}), **opts), ticks)
    return _FOUR[key]


def _result_of(s, rank=0, pipelined=False):
    res = dict(rank=rank, merges=s.merges, hosted=s.hosted(), refinements=s.refinements, frame_of=s.frame_of,
               stats=s.async_stats() if pipelined else None)
    if s.hosted():
        frames = sorted({s.frame_of[c] for c in s.hosted()})
        res.update(maps={f: s.cams[f].model() for f in frames}, fern_frames={f: len(s.ferns[f]) for f in frames},
                   pose_graph={c: s.pose_graph[c] for c in s.hosted()}, relative_cons={c: s.relative_cons[c] for c in s.hosted()})
    return res


@pytest.mark.parametrize("impl,pipelined", [("native", False), ("native", True), ("python", False)])
def test_four_cameras_one_device_end_in_one_map(orc, impl, pipelined):
    """--n 4 on one device: frame 0 consumes camera 1's map, frame 2 consumes frame 0 (two cameras), frame 3 consumes frame 2 (three
    cameras: its founder and two that had joined) - ONE map with four time slots, and all four cameras keep tracking against and fusing
    into it.  Synchronous tick (merges at ticks 6, 7, 8) and pipelined tick (each merge three ticks after its descriptor hit).  Map,
    four trajectories, three transforms, key frames and the re-based constraint rows: the one-process oracle session's bits."""
    from densemonoslam_amd import synth

    ref, ticks = four_oracle(3 if pipelined else None)
    sc = SCENARIOS["reference_rule"]
    s = _make_session(impl, sc, 4, capacity=4_000_000)
    for k in range(ticks):
        s.step(k, frames_at(synth, k, FOUR_OFFSETS), **({"pipelined": True} if pipelined else {}))
    res = _result_of(s, 0, pipelined)
    if pipelined:
        assert res["stats"] == {"ticks": ticks, "woken": len(ref.woken)}
    check_four(ref, {0: res}, 1, ticks)
    s.close()


def _worker_n(rank, world, port, q, impl, ticks, offsets, pipelined, opts=None):
    sc = SCENARIOS["reference_rule"]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["DMS_TRACK_MODE"] = "launches"  # processes that share a device must not spin side by side
    import copy

    import torch.distributed as dist

    from densemonoslam_amd import synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc2 = copy.copy(sc)
    sc2.opts = dict(sc.opts, **(opts or {}))
    s = _make_session(impl, sc2, len(offsets), rank, world, capacity=4_000_000)
    for k in range(ticks):
        fr = frames_at(synth, k, offsets)
        s.step(k, {c: fr[c] for c in fr if c % world == rank}, **({"pipelined": True} if pipelined else {}))
    q.put(_result_of(s, rank, pipelined))
    dist.barrier()
    s.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
def test_four_cameras_two_ranks_end_in_one_map(orc, pipelined):
    """--n 4 over two ranks (gloo transport, one GPU): cameras 0 and 2 are read on rank 0, 1 and 3 on rank 1.  The first merge crosses
    ranks (records, import, frame forwarding), the second joins on rank 0, the third ships a frame that holds three cameras - a founder, an
    IMPORTED camera and a joined one - to rank 1, which ends up hosting all four while rank 0 only forwards frames."""
    world = 2
    ref, ticks = four_oracle(3 if pipelined else None)
    results = spawn(world, _worker_n, ("native", ticks, FOUR_OFFSETS, pipelined))
    if pipelined:
        for r in range(world):
            assert results[r]["stats"] == {"ticks": ticks, "woken": len(ref.woken)}
    check_four(ref, results, world, ticks)


def test_queries_inside_the_frame_as_the_reference_orders_them(orc):
    """dms_session_params.query_inside_frame: the inter-map block runs INSIDE each camera's processFrame (ElasticFusion.cpp:595-632) -
    camera c queries, and merges, before camera c + 1's frame of the same tick.  Four cameras on one device: at tick 6 cameras 1 AND 2
    are consumed by frame 0 in the same tick (the batched order allows one merge per frame and tick: ticks 6, 7, 8 there), and camera 3's
    frame of that tick... stays in its own map until camera 0 finds it.  Against orc_pipeline.Session(query_inside_frame = True); and the
    two orders really differ."""
    from densemonoslam_amd import synth

    ticks = 11
    ref = run_oracle_session_n(FOUR_OFFSETS, ticks, query_inside_frame=True)
    batched, _ = four_oracle(None)
    assert [(m[0], m[1], m[2]) for m in ref.merges] != [(m[0], m[1], m[2]) for m in batched.merges]
    assert len({m[0] for m in ref.merges}) < len(ref.merges), "two merges in one tick are what the order makes possible"
    import copy

    sc = copy.copy(SCENARIOS["reference_rule"])
    sc.opts = dict(sc.opts, query_inside_frame=True)
    s = _make_session("native", sc, 4, capacity=4_000_000)
    for k in range(ticks):
        s.step(k, frames_at(synth, k, FOUR_OFFSETS))
    check_four(ref, {0: _result_of(s)}, 1, ticks)
    s.close()


def test_queries_inside_the_frame_two_ranks(orc):
    """the same order across two ranks: per camera one published block, one table, one walk (the cameras of a tick are serialised)"""
    ticks = 11
    ref = run_oracle_session_n(FOUR_OFFSETS, ticks, query_inside_frame=True)
    results = spawn(2, _worker_n, ("native", ticks, FOUR_OFFSETS, False, dict(query_inside_frame=True)))
    check_four(ref, results, 2, ticks)


def test_pipelined_session_over_a_one_rank_rccl_communicator(pipelined_oracle):
    """The pipelined tick with the RCCL transport (what `bench.py --gpus N` times): with a transport even a one-rank session takes the
    multi-rank arrangement - every map's frames on a stream of its own, the exchange (key-frame insertion, a real ncclAllGather of the frame
    blocks, descriptor search, host mirror) on the caller's stream beside the next frames, the all-gather bracketed by HIP events
    (time_exchange).  Same wake, merge and bits as the oracle session on that schedule."""
    from densemonoslam_amd import capi, collab, session, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    ref = pipelined_oracle
    sc = SCENARIOS["reference_rule"]
    tr = session.RcclTransport(collab.RcclCarrier(0, 1, collab.RcclCarrier.unique_id()))
    assert "rccl" in tr.carrier.library_path()
    s = session.NativeSession(W, H, K, 2, rank=0, world=1, transport=tr, fern_photo_thresh=sc.fern_photo, model_capacity=2_000_000, time_exchange=True,
                              **sc.opts)
    row = lambda c: np.arange(6, dtype=np.float32) * np.float32(0.25 + c)
    for c in s.hosted():
        s.addRelativeConstraint(c, row(c)[:3], row(c)[3:])
    for k in range(PIPE_TICKS):
        s.step(k, sc.frames(synth, k), pipelined=True)
    s.sync()
    assert s.async_stats() == {"ticks": PIPE_TICKS, "woken": len(ref.woken)}
    ms, n = s.exchange_time()
    assert n >= PIPE_TICKS - 3 and 0.0 < ms / n < 5.0, (ms, n)  # (the all-gathers of the ticks whose completion the host has seen; a one-rank gather is a copy)
    fb = ref.merges[0][1]
    host = dict(map=s.cams[fb].model(), fern_frames=len(s.ferns[fb]), pose_graph=s.pose_graph, relative_cons=s.relative_cons)
    _check(ref, host, s.merges, PIPE_TICKS)
    s.close()
