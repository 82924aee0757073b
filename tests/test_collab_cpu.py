"""World-size-2 gloo test of the collaborative (N > 1) path on CPU: camera sharding, the
per-frame thumbnail all-gather and the max-over-ranks timing reduction used by bench.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _thumbs_for_camera(cam, W, H):
    """Thumbnail block of camera `cam` built on the CPU from the synthetic stream (oracle resize)."""
    from densemonoslam_amd import synth
    from oracle import orc

    K = (0.825 * W, 0.825 * W, W / 2.0, H / 2.0)
    d, rgb, _ = synth.frame(3, cam_id=cam, width=W, height=H, K=K, noise=False)
    rgba = synth.rgba(rgb)
    vmap = orc.createVMap(K, d, 20.0)
    v4 = np.zeros((H, W, 4), np.float32)
    for c in range(3):
        v4[..., c] = np.nan_to_num(vmap[c * H:(c + 1) * H])
    n4 = np.roll(v4, 1, axis=2)
    parts = [orc.resize_nn(rgba, H // 8, W // 8), orc.resize_nn(v4, H // 8, W // 8), orc.resize_nn(n4, H // 8, W // 8)]
    return np.concatenate([p.reshape(-1).view(np.uint8) for p in parts])


def _worker(rank, world, port, W, H, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from densemonoslam_amd import collab

    dev = torch.device("cpu")
    cams = collab.shard_cameras(world, rank, world)
    assert cams == [rank]
    ex = collab.ThumbnailExchange(world, W, H, dev)
    block = _thumbs_for_camera(cams[0], W, H)
    assert block.size == ex.nbytes == collab.thumbnail_bytes(W, H)
    ex.local.copy_(torch.from_numpy(block))
    g = ex.gather().numpy().copy()
    tmax = collab.max_over_ranks(0.5 + rank, dev)
    tsum = collab.sum_over_ranks(10 + rank, dev)
    out.put((rank, g, tmax, tsum))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_thumbnail_allgather_gloo():
    world, W, H = 2, 160, 120
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.stack([_thumbs_for_camera(c, W, H) for c in range(world)])
    assert (expect[0] != expect[1]).any(), "the two cameras must see different thumbnails"
    for rank, g, tmax, tsum in results:
        assert g.shape == expect.shape and (g == expect).all(), "rank %d gathered wrong thumbnails" % rank
        assert tmax == 1.5 and tsum == 21.0


def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from densemonoslam_amd import collab

    ex = collab.ThumbnailExchange(world, 160, 120, torch.device("cpu"))
    got = []
    handles = []
    for frame in range(5):  # the block of frame t is refilled at frame t+2, after its collective was waited for
        buf = ex.begin()
        buf.fill_(10 * frame + rank)
        handles.append(ex.gather(overlap=True))
        if frame >= 1:  # the matcher works one frame behind: wait for the previous frame's gather only
            ex.work[ex.slot].wait() if ex.work[ex.slot] is not None else None
            got.append(handles[frame - 1][:, 0].clone().numpy())
    ex.finish()
    got.append(handles[4][:, 0].clone().numpy())
    out.put((rank, np.stack(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_thumbnail_exchange_gloo():
    """bench.py's use of the exchange: double-buffered blocks, all-gather started without waiting and
    consumed a frame later."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.array([[10 * f + r for r in range(world)] for f in range(5)], np.uint8)
    for rank, got in results:
        assert (got == expect).all(), (rank, got)


def _merge_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from densemonoslam_amd import collab
    from oracle import orc

    rng = np.random.default_rng(100 + rank)
    n = 1000 + 500 * rank
    m = np.zeros(n, orc.SURFEL_DTYPE)
    for k in ("pos", "col", "nrm", "times"):
        m[k] = rng.normal(size=m[k].shape).astype(np.float32)
    rec = torch.from_numpy(m.view(np.float32).reshape(n, collab.RECORD_FLOATS).copy())
    if rank == 1:  # rank 1's map is consumed by rank 0
        collab.send_map(rec, n, 0)
        out.put((rank, m, None))
    else:
        got, count = collab.recv_map(1, torch.device("cpu"))
        theirs = got[:count].numpy().copy().view(orc.SURFEL_DTYPE).reshape(count)
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = (1.0, 2.0, 3.0)
        out.put((rank, m, orc.model_consume(m, theirs, T)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_map_merge_transfer_gloo():
    """The map-merge transfer of the collaborative mode: rank 1 sends its map as packed records, rank 0
    receives it and consumes it (oracle consume on the CPU; the HIP consume is tested on the GPU)."""
    from oracle import orc

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, own, merged = q.get(timeout=180)
        res[rank] = (own, merged)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    own0, merged = res[0]
    own1, _ = res[1]
    assert len(merged) == len(own0) + len(own1)
    assert (merged[:len(own0)].view(np.uint32) == own0.view(np.uint32)).all()
    tail = merged[len(own0):]
    assert np.array_equal(tail["col"], own1["col"]) and np.array_equal(tail["times"], own1["times"])
    assert np.allclose(tail["pos"][:, :3], own1["pos"][:, :3] + np.array([1, 2, 3], np.float32), atol=1e-6)
    assert np.array_equal(tail["pos"][:, 3], own1["pos"][:, 3]) and np.array_equal(tail["nrm"], own1["nrm"])


def test_camera_sharding():
    from densemonoslam_amd import collab

    assert collab.shard_cameras(8, 3, 8) == [3]
    assert collab.shard_cameras(4, 1, 2) == [1, 3]
    assert sorted(sum((collab.shard_cameras(5, r, 2) for r in range(2)), [])) == [0, 1, 2, 3, 4]
    assert collab.thumbnail_bytes(640, 480) == 80 * 60 * 36


# ---- the inter-map matcher's per-frame protocol, world size 2 on gloo ----------------------------------------
class _CpuCamera:
    """stand-in for fusion.ElasticFusion on a CPU rank: hands out this camera's thumbnails and pose"""

    def __init__(self, cam, W, H):
        import ctypes

        self.ct, self.cam, self.W, self.H = ctypes, cam, W, H
        self.pose = np.eye(4, dtype=np.float32)
        self.pose[:3, 3] = (cam, 0.5, -0.25)
        self.block = _thumbs_for_camera(cam, W, H)

    def thumbnails(self, ptr, stream):
        self.ct.memmove(ptr, self.block.ctypes.data, self.block.nbytes)

    def poseDevice(self):
        return self.pose.ctypes.data

    def exportPose(self, ptr, stream):
        self.ct.memmove(ptr, self.pose.ctypes.data, 64)


class _CpuFerns:
    """stand-in for ferns.Ferns built on the CPU restatement (oracle/orc_ferns.py), same call surface"""

    def __init__(self, W, H, seed):
        import ctypes
        from oracle import orc_ferns

        self.ct = ctypes
        self.o = orc_ferns.Ferns(W, H, (0.825 * W, 0.825 * W, W / 2.0, H / 2.0), seed=seed)
        self.tw, self.th = W // 8, H // 8
        self.searches = []

    def _unpack(self, ptr):
        n = self.tw * self.th
        raw = np.ctypeslib.as_array((self.ct.c_ubyte * (n * 36)).from_address(ptr)).copy()
        return (raw[:n * 4].reshape(self.th, self.tw, 4), raw[n * 4:n * 20].view(np.float32).reshape(self.th, self.tw, 4),
                raw[n * 20:].view(np.float32).reshape(self.th, self.tw, 4))

    def encodeThumbs(self, ptr, codes_ptr, good_ptr, stream):
        img, v, _ = self._unpack(ptr)
        codes, good, _ = self.o._encode(img, v)
        full = np.full(512, 255, np.uint8)
        full[:len(codes)] = codes
        self.ct.memmove(codes_ptr, full.ctypes.data, 512)
        self.ct.memmove(good_ptr, np.array([good], np.int32).ctypes.data, 4)

    def addFrameAsync(self, ptr, pose_ptr, tick, thr, stream):
        img, v, n = self._unpack(ptr)
        pose = np.ctypeslib.as_array((self.ct.c_float * 16).from_address(pose_ptr)).copy()
        self.o._add(img, v, n, pose, tick, thr)

    def searchCodes(self, codes_ptr, good_ptr, tick, interMap, best_ptr, stream):
        codes = np.ctypeslib.as_array((self.ct.c_ubyte * 512).from_address(codes_ptr)).copy()[:self.o.num]
        good = int(np.ctypeslib.as_array((self.ct.c_int * 1).from_address(good_ptr))[0])
        best, bid = np.float32(3.4e38), -1
        for i, fr in enumerate(self.o.frames):
            co = int(((codes == fr.codes) & (codes != 255)).sum())
            m = np.float32(min(good, fr.goodCodes))
            d = np.float32(m - np.float32(co)) / m
            if d < best:
                best, bid = d, i
        out = np.array([bid, np.array([best], np.float32).view(np.int32)[0] if bid >= 0 else -1], np.int32)
        self.ct.memmove(best_ptr, out.ctypes.data, 8)
        self.searches.append((tick, bid, float(best)))


def _matcher_worker(rank, world, port, W, H, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from densemonoslam_amd import collab

    dev = torch.device("cpu")
    ex = collab.ThumbnailExchange(world, W, H, dev, extra_bytes=collab.DESC_BYTES)
    cam, ferns = _CpuCamera(rank, W, H), _CpuFerns(W, H, seed=99)
    m = collab.InterMapMatcher(ferns, ex, rank, world, dev, fern_threshold=0.3)
    seen = []
    for tick in range(1, 5):
        prev = m.publish(cam, tick, None)
        m.match(prev, tick, None)
        if prev is not None:
            g = prev[0]
            T = ex.thumb_bytes
            other = 1 - rank
            seen.append((int(g[other][T + collab.DESC_TICK:T + collab.DESC_TICK + 4].view(torch.int32)[0]),
                         g[other][T + collab.DESC_POSE:T + collab.DESC_POSE + 64].view(torch.float32).numpy().copy(),
                         int(g[other][T + collab.DESC_GOOD:T + collab.DESC_GOOD + 4].view(torch.int32)[0])))
    ex.finish()
    out.put((rank, seen, ferns.searches, len(ferns.o.frames), m.best_host.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_intermap_matcher_protocol_gloo():
    """Descriptor + thumbnails of each camera reach the other rank one frame later (tick, pose, good-code count intact),
    every rank searches its own fern database with the remote descriptor, and the own frame is offered to the local
    database each frame.  Both cameras see the same synthetic room from different places."""
    world, W, H = 2, 320, 240
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_matcher_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r[0]: r for r in [q.get(timeout=240) for _ in range(world)]}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        _, seen, searches, nframes, best = results[rank]
        other = 1 - rank
        assert [s[0] for s in seen] == [1, 2, 3]  # the remote descriptor of tick t is consumed at tick t + 1
        for tick, pose, good in seen:
            assert pose.reshape(4, 4)[0, 3] == float(other) and good > 0
        assert nframes == 1  # the static camera's first frame is a key frame, the identical later ones are rejected
        assert len(searches) == 3 and all(s[1] == 0 for s in searches)  # one stored frame: it is the candidate
        assert best[other, 0] == 0 and best[rank, 0] == -1


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher around it must run TWO ranks (round 2: the flag was parsed and ignored, the
    job ran one rank and reported n_gpus = 1).  Rendezvous rehearsal without a GPU (DMS_BENCH_DRY=1: the ranks meet over gloo and
    rank 0 reports the job's shape); a mismatch between --gpus and the launcher's world size must fail loudly."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, DMS_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and sorted(x["rank"] for x in d["ranks"]) == [0, 1] and sorted(x["local_rank"] for x in d["ranks"]) == [0, 1]
    # one rank, no launcher
    r = subprocess.run([sys.executable, bench, "--gpus", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # a launcher's world size that contradicts --gpus
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_collab_c_binding_validates_arguments_without_a_gpu():
    """include/dmslam_collab.h: the entry points exist, refuse bad arguments before touching RCCL, and a single-camera
    process has not loaded librccl by importing the package."""
    import ctypes as C

    from densemonoslam_amd import capi

    lib = capi.lib
    assert lib.dms_collab_unique_id(None) == -1
    h = C.c_void_p()
    ident = b"\0" * 128
    assert lib.dms_collab_create(C.byref(h), 2, 2, ident) == -1 and b"rank" in lib.dms_last_error()
    assert lib.dms_collab_create(C.byref(h), 0, 0, ident) == -1
    assert lib.dms_collab_create(None, 0, 1, ident) == -1
    assert lib.dms_collab_destroy(None) == 0
    lib.dms_collab_size.argtypes = [C.c_void_p]
    assert lib.dms_collab_size(None) == 0
    maps = open("/proc/self/maps").read()
    assert "libdmslam_hip" in maps
