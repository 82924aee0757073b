"""The library's RCCL binding (include/dmslam_collab.h) on a real device: a one-rank communicator carries the per-frame
all-gather and the timing all-reduce (RCCL executes; two ranks cannot share this box's single GPU — the N-rank path is
covered by the gloo tests of tests/test_collab_cpu.py and run by `bench.py --gpus N`), and the frame step's exchange through
it is the same as through torch.distributed's local copy."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def carrier():
    import torch

    from densemonoslam_amd import collab

    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # the box has no network: the bootstrap listens on loopback
    torch.cuda.set_device(0)
    c = collab.RcclCarrier(0, 1, collab.RcclCarrier.unique_id())
    yield c
    c.close()


def test_one_rank_communicator_moves_bytes(carrier):
    import torch

    assert (carrier.rank, carrier.world) == (0, 1)
    dev = torch.device("cuda", 0)
    src = torch.randint(0, 256, (173392,), dtype=torch.uint8, device=dev)  # the 640x480 frame block
    dst = torch.zeros_like(src)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    carrier.allgather(src, dst, s)
    v = torch.tensor([3.25], dtype=torch.float64, device=dev)
    carrier.max_f64(v, s)
    s.synchronize()
    assert torch.equal(src, dst) and float(v[0]) == 3.25
    from densemonoslam_amd import capi

    with pytest.raises(capi.DmsError):  # a rank does not send to itself
        carrier.send(src, 0, s)


def test_matcher_through_the_binding_equals_the_local_copy(carrier):
    """InterMapMatcher's per-frame protocol (publish, all-gather, search a frame later) with the library's collective in
    place of the world-size-1 local copy: same gathered blocks, same key-frame database."""
    import torch

    from densemonoslam_amd import collab, ferns as ferns_mod, fusion, synth

    W, H, K = 320, 240, (264.0, 264.0, 160.0, 120.0)
    dev = torch.device("cuda", 0)
    res = []
    for use in (None, carrier):
        ef = fusion.ElasticFusion(W, H, K)
        db = ferns_mod.Ferns(W, H, K, num=500, maxDepth_mm=3000, photoThresh=115.0, seed=7, capacity=256)
        x = collab.ThumbnailExchange(1, W, H, dev, extra_bytes=collab.DESC_BYTES, carrier=use)
        m = collab.InterMapMatcher(db, x, 0, 1, dev)
        stream = torch.cuda.current_stream().cuda_stream
        blocks = []
        prev = None
        for k in range(6):
            d, rgb, _ = synth.frame(k, width=W, height=H, K=K, noise=True)
            ef.processFrame(rgb, d)
            m.match(prev, k, stream)
            prev = m.publish(ef, k, stream)
            x.finish()
            torch.cuda.synchronize()
            blocks.append(x.gathered.cpu().numpy().copy())
        res.append((blocks, len(db)))
    for a, b in zip(res[0][0], res[1][0]):
        assert a.tobytes() == b.tobytes()
    assert res[0][1] == res[1][1] >= 1
