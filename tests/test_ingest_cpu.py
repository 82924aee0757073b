"""Frame ingest formats: the C-ABI codec / readers against golden vectors produced by
the reference's own eflcm.Frame class (tests/golden/make_eflcm_golden.py), and against the oracle's
restatement for the two log containers.  No GPU needed."""
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "eflcm_frames.npz"))


@pytest.fixture(scope="module")
def ing():
    from densemonoslam_amd import ingest

    return ingest


def _case(g, i):
    return dict(enc=g["enc%d" % i].tobytes(), depth=g["depth%d" % i].tobytes(), image=g["image%d" % i].tobytes(), flags=g["flags%d" % i],
                stamp=g["stamp%d" % i], sender=g["sender%d" % i].tobytes().decode("utf-8"))


@pytest.mark.parametrize("i", [0, 1, 2])
def test_frame_codec_matches_reference_bytes(ing, golden, i):
    from oracle import orc_frame

    c = _case(golden, i)
    # decode the reference's bytes
    f = ing.Frame.decode(c["enc"])
    assert (f.trackOnly, f.compressed, f.last) == tuple(bool(x) for x in c["flags"])
    assert f.depth == c["depth"] and f.image == c["image"]
    assert (f.timestamp, f.frameNumber) == tuple(int(x) for x in c["stamp"])
    assert f.senderName == c["sender"]
    # encode the same fields: byte-identical to Frame.encode()
    g = ing.Frame(c["depth"], c["image"], int(c["stamp"][0]), int(c["stamp"][1]), c["sender"], *[bool(x) for x in c["flags"]])
    assert g.encode() == c["enc"]
    # the oracle restatement is pinned by the same vectors
    assert orc_frame.frame_encode(bool(c["flags"][0]), bool(c["flags"][1]), bool(c["flags"][2]), c["depth"], c["image"], int(c["stamp"][0]),
                                  int(c["stamp"][1]), c["sender"]) == c["enc"]
    d = orc_frame.frame_decode(c["enc"])
    assert d["depth"] == c["depth"] and d["senderName"] == c["sender"] and d["timestamp"] == int(c["stamp"][0])
    assert golden["fingerprint"].tobytes() == orc_frame.FINGERPRINT


def test_frame_decode_rejects_garbage(ing, golden):
    from densemonoslam_amd import capi

    enc = _case(golden, 0)["enc"]
    with pytest.raises(capi.DmsError):
        ing.Frame.decode(b"\0" * 8 + enc[8:])  # wrong fingerprint ("Decode error" in the reference)
    with pytest.raises(capi.DmsError):
        ing.Frame.decode(enc[:40])  # truncated payload


def test_frame_unpack_raw_and_compressed(ing, golden):
    from densemonoslam_amd import capi

    W, H = (int(x) for x in golden["shape"])
    c = _case(golden, 0)
    f = ing.Frame.decode(c["enc"])
    d, rgb = f.unpack(W, H)
    assert d.tobytes() == c["depth"] and rgb.tobytes() == c["image"]
    d2, rgb2 = f.unpack(W, H, flipColors=True)
    assert (rgb2[..., 0] == rgb[..., 2]).all() and (rgb2[..., 2] == rgb[..., 0]).all() and (rgb2[..., 1] == rgb[..., 1]).all()
    # compressed message: zlib depth decodes; its 41 opaque image bytes are not a JPEG stream
    f1 = ing.Frame.decode(_case(golden, 1)["enc"])
    assert f1.compressed and f1.last and f1.trackOnly
    with pytest.raises(capi.DmsError) as e:
        f1.unpack(W, H)
    assert "JPEG" in str(e.value)
    assert np.frombuffer(zlib.decompress(f1.depth), np.uint16).tolist() == golden["depth1_raw"].tolist()


def test_lcm_log_reader(ing, golden, tmp_path):
    from oracle import orc_frame

    path = str(tmp_path / "log.lcm")
    msgs = [_case(golden, i)["enc"] for i in (0, 2, 1)]
    orc_frame.lcmlog_write(path, [(1000 * (k + 1), "EFUSION_FRAMES" if k != 1 else "OTHER", m) for k, m in enumerate(msgs)])
    r = ing.LcmLogReader(path)
    ev = list(r)
    assert [(c, t) for c, _, t in ev] == [("EFUSION_FRAMES", 1000), ("OTHER", 2000), ("EFUSION_FRAMES", 3000)]
    assert [d for _, d, _ in ev] == msgs
    assert ing.Frame.decode(ev[2][1]).last  # RawLcmLogReader stops at f.last
    r.rewind()
    assert len(list(r)) == 3
    r.close()


@pytest.mark.parametrize("compress", [False, True])
def test_klg_reader(ing, tmp_path, compress):
    from oracle import orc_frame

    W, H = 16, 12
    rng = np.random.default_rng(3)
    frames = [(10 + k, rng.integers(0, 6000, (H, W), dtype=np.uint16), None if k == 1 else rng.integers(0, 256, (H, W, 3), dtype=np.uint8))
              for k in range(3)]
    path = str(tmp_path / "log.klg")
    orc_frame.klg_write(path, frames, compress_depth=compress)
    r = ing.KlgReader(path, W, H)
    assert r.numFrames == 3
    got = list(r)
    assert len(got) == 3
    for (ts, d, rgb), (ts0, d0, rgb0) in zip(got, frames):
        assert ts == ts0 and (d == d0).all()
        assert (rgb == (rgb0 if rgb0 is not None else 0)).all()  # imageSize 0 => black image (RawLogReader.cpp:106-110)
    r.rewind()
    assert next(iter(r))[0] == 10
    r.close()


# ---- JPEG colour (GUI/src/Tools/JPEGLoader.h): golden vectors encoded and decoded by libjpeg-turbo (make_jpeg_golden.py) ----
@pytest.fixture(scope="module")
def jpeg_cases():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_cases.npz"))


def test_jpeg_decoder_reproduces_libjpeg_bytes(ing, jpeg_cases):
    import hashlib

    z = jpeg_cases
    assert len(z["names"]) >= 12
    for name in z["names"]:
        name = str(name)
        w, h = (int(v) for v in z[name + "_shape"])
        got = ing.jpeg_decode(z[name + "_jpeg"].tobytes(), w, h)
        ref = z[name + "_rgb"]
        assert (got[: ref.shape[0]] == ref).all(), name
        assert hashlib.sha256(got.tobytes()).digest() == z[name + "_sha256"].tobytes(), name


def test_jpeg_decoder_refuses_what_it_cannot_decode(ing, jpeg_cases):
    from densemonoslam_amd import capi

    z = jpeg_cases
    good = z["q95_420_jpeg"].tobytes()
    with pytest.raises(capi.DmsError) as e:
        ing.jpeg_decode(z["progressive_jpeg"].tobytes(), 32, 24)
    assert e.value.code == ing.DMS_ERR_UNSUPPORTED and "progressive" in str(e.value)
    with pytest.raises(capi.DmsError) as e:
        ing.jpeg_decode(good, 32, 24)  # not the log's resolution
    assert e.value.code == ing.DMS_ERR_FORMAT
    for cut in (2, 100, len(good) // 2):
        try:  # a truncated stream is either refused or decoded with libjpeg's zero-fill rule; it must not crash or read past the end
            ing.jpeg_decode(good[:cut], 64, 48)
        except capi.DmsError as err:
            assert err.code == ing.DMS_ERR_FORMAT
    with pytest.raises(capi.DmsError):
        ing.jpeg_decode(b"\x00" * 64, 64, 48)
    rng = np.random.default_rng(5)
    for _ in range(200):  # corrupted entropy-coded data: any outcome but a crash
        bad = bytearray(good)
        for k in rng.integers(300, len(bad), 4):
            bad[k] = int(rng.integers(0, 256))
        try:
            ing.jpeg_decode(bytes(bad), 64, 48)
        except capi.DmsError:
            pass


def test_compressed_logs_decode_like_the_reference_readers(ing, jpeg_cases, tmp_path):
    """RawLogReader (.klg) and RawLcmLogReader (eflcm.Frame, compressed = 1): zlib depth + JPEG colour; the loader leaves the
    colour bytes as B, G, R (JPEGLoader.h:82-91: libjpeg's scanline with R and B exchanged), flipColors exchanges them again."""
    import struct

    z = jpeg_cases
    W, H = (int(v) for v in z["q95_420_shape"])
    jpg = z["q95_420_jpeg"].tobytes()
    rgb = z["q95_420_rgb"]
    rng = np.random.default_rng(9)
    depth = rng.integers(0, 6000, (H, W), dtype=np.uint16)
    zd = zlib.compress(depth.tobytes(), 6)
    path = str(tmp_path / "c.klg")
    with open(path, "wb") as f:
        f.write(struct.pack("<i", 2))
        for ts in (5, 6):
            f.write(struct.pack("<qii", ts, len(zd), len(jpg)) + zd + jpg)
    for flip in (False, True):
        r = ing.KlgReader(path, W, H, flipColors=flip)
        got = list(r)
        r.close()
        assert [g[0] for g in got] == [5, 6]
        for _, d, c in got:
            assert (d == depth).all()
            assert (c == (rgb if flip else rgb[..., ::-1])).all()
    f = ing.Frame(zd, jpg, 77, 3, "cam", False, True, False)
    g = ing.Frame.decode(f.encode())
    d, c = g.unpack(W, H)
    assert (d == depth).all() and (c == rgb[..., ::-1]).all()
    d, c = g.unpack(W, H, flipColors=True)
    assert (c == rgb).all()
