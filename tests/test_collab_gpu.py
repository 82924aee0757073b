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


def test_a_rank_that_never_arrives_is_an_error_not_a_hang():
    """dms_collab_create for rank 0 of 2 whose peer never calls it: ncclCommInitRank would block for ever; the library runs it on a helper
    thread and returns DMS_ERR_TIMEOUT after DMS_RCCL_INIT_TIMEOUT_S.  In a child process (the stuck helper thread ends with it)."""
    import os
    import subprocess
    import sys
    import time

    code = (
        "import sys, time, ctypes as C\n"
        "import torch\n"
        "from densemonoslam_amd import capi, collab\n"
        "assert capi.device_count() >= 1\n"
        "t0 = time.time()\n"
        "try:\n"
        "    collab.RcclCarrier(0, 2, collab.RcclCarrier.unique_id())\n"
        "    print('CREATED')\n"
        "except Exception as e:\n"
        "    print('ERR %.1f %s' % (time.time() - t0, e))\n"
        "sys.stdout.flush()\n"
        "import os\n"
        "os._exit(0)\n")
    env = dict(os.environ, DMS_RCCL_INIT_TIMEOUT_S="4", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180)
    out = r.stdout.strip().splitlines()
    line = [l for l in out if l.startswith("ERR") or l.startswith("CREATED")]
    assert line and line[-1].startswith("ERR"), (r.stdout[-400:], r.stderr[-400:])
    assert "did not return within" in line[-1] and time.time() - t0 < 120, line[-1]
