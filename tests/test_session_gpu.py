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

from tests.test_session_cpu import FERN_PHOTO, H, K, OFFSET, QUERY_FROM, W, _free_port, frames_for

pytestmark = pytest.mark.gpu
N_TICKS = QUERY_FROM + 21
SESSION_OPTS = dict(inter_map=2, query_from=QUERY_FROM)


@pytest.fixture(scope="module")
def oracle_session(orc):
    from densemonoslam_amd import synth
    from oracle import orc_pipeline

    s = orc_pipeline.Session(2, W, H, K, fern_photo_thresh=FERN_PHOTO, **SESSION_OPTS)
    for k in range(N_TICKS):
        fr = frames_for(synth, k)
        s.step([fr[0], fr[1]], k)
    assert len(s.merges) == 1 and N_TICKS - 1 - s.merges[0][0] >= 20, s.merges
    return s


def _check(ref, host, merges):
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


def test_two_cameras_one_device_merge_and_continue(oracle_session):
    import torch

    from densemonoslam_amd import capi, session, synth

    assert capi.device_count() >= 1, "no MI355X visible"
    be = session.GpuBackend(W, H, K, torch.device("cuda", 0), fern_opts=dict(photoThresh=FERN_PHOTO), model_capacity=2_000_000)
    s = session.CollabSession(be, 2, W, H, **SESSION_OPTS)
    for k in range(N_TICKS):
        s.step(k, frames_for(synth, k))
    fb = oracle_session.merges[0][1]
    host = dict(map=s.cams[fb].model(), fern_frames=len(s.ferns[fb]), pose_graph=s.pose_graph)
    _check(oracle_session, host, s.merges)
    s.close()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["DMS_TRACK_MODE"] = "launches"  # two processes on one device must not spin side by side
    import torch
    import torch.distributed as dist

    from densemonoslam_amd import session, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = session.GpuBackend(W, H, K, torch.device("cuda", 0), fern_opts=dict(photoThresh=FERN_PHOTO), model_capacity=2_000_000)
    s = session.CollabSession(be, 2, W, H, rank=rank, world=world, **SESSION_OPTS)
    for k in range(N_TICKS):
        fr = frames_for(synth, k)
        s.step(k, {c: fr[c] for c in fr if c % world == rank})
    res = dict(rank=rank, merges=s.merges, hosted=s.hosted())
    if s.hosted():
        fb = s.frame_of[s.hosted()[0]]
        res.update(map=s.cams[fb].model(), fern_frames=len(s.ferns[fb]), pose_graph={c: s.pose_graph[c] for c in s.hosted()})
    q.put(res)
    dist.barrier()
    s.close()
    dist.destroy_process_group()


def test_two_ranks_merge_across_processes_and_continue(oracle_session):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
    _check(oracle_session, results[hb], results[hb]["merges"])
