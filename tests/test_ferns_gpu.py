"""Fern keyframe database (Ferns.cpp) — HIP path through the C ABI against the CPU restatement
(oracle/orc_ferns.py).  Integer work (table, codes, good-code counts, the add / reject decisions, candidate ids) must be
exact; the dissimilarity is one float division (exact); the verified pose follows the tracker's bar (<= 1 mm / 0.01 deg)."""
import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

W, H = 640, 480  # thumbnails 80 x 60, as in the reference's configuration
K = (528.0, 528.0, 320.0, 240.0)


@pytest.fixture(scope="module")
def mods():
    from densemonoslam_amd import capi, ferns, fusion, synth
    from oracle import orc, orc_ferns

    assert capi.device_count() >= 1
    return ferns, fusion, synth, orc, orc_ferns


def textures(fusion, synth, k, noise=True):
    """fill-in-like textures of synthetic frame k: RGBA8 image, RGBA32F vertex / normal maps (camera frame) + pose"""
    from oracle import orc

    d, rgb, T = synth.frame(k, width=W, height=H, K=K, noise=noise)
    vo = orc.createVMap(K, d, 3.0)
    no = orc.createNMap(vo)
    ok = ~np.isnan(vo[:H]) & ~np.isnan(no[:H])
    v = np.zeros((H, W, 4), np.float32)
    n = np.zeros((H, W, 4), np.float32)
    for c in range(3):
        v[..., c] = np.where(ok, vo[c * H:(c + 1) * H], 0)
        n[..., c] = np.where(ok, no[c * H:(c + 1) * H], 0)
    return synth.rgba(rgb), v, n, T.astype(np.float32)


def make_pair(mods, seed=11, **kw):
    ferns, fusion, synth, orc, orc_ferns = mods
    g = ferns.Ferns(W, H, K, seed=seed, **kw)
    o = orc_ferns.Ferns(W, H, K, seed=seed, make_odometry=lambda: orc.Odometry(W // 8, H // 8, K[2] / 8, K[3] / 8, K[0] / 8, K[1] / 8), **kw)
    return g, o


def test_table_and_codes_exact(mods):
    ferns, fusion, synth, orc, orc_ferns = mods
    g, o = make_pair(mods)
    pos, rgbd = g.table()
    assert (pos == o.pos).all() and (rgbd == o.rgbd).all()
    assert pos[:, 0].max() < W // 8 and pos[:, 1].max() < H // 8 and rgbd[:, 3].min() >= 400 and rgbd[:, 3].max() <= 3000
    for k in (0, 7):
        img, v, n, _ = textures(fusion, synth, k)
        if k == 7:  # a hole: bad codes
            v[60:180, 80:240] = 0
        codes, good = g.encode(img, v, n)
        ti, tv, tn = o._thumbs(img, v, n)
        co, go, _ = o._encode(ti, tv)
        assert good == go and (codes == co).all()
        assert (codes == 255).sum() == 500 - good
    g.close()


def test_add_frame_decisions_and_database_exact(mods):
    """A walk through the synthetic room: every accept / reject decision of addFrame, the stored codes, good counts,
    times and poses must equal the reference loop's (inverted co-occurrence lists) exactly."""
    ferns, fusion, synth, orc, orc_ferns = mods
    g, o = make_pair(mods)
    decisions = []
    for k in range(0, 40, 2):
        img, v, n, T = textures(fusion, synth, k)
        a = g.addFrame(img, v, n, T, k, 0.3)
        b = o.addFrame(img, v, n, T, k, 0.3)
        assert a == b, k
        decisions.append(a)
    assert decisions[0] and 1 < sum(decisions) < len(decisions)  # first frame always; some rejected as too similar
    assert len(g) == len(o.frames)
    for i, fr in enumerate(o.frames):
        pose, t, good, codes = g.frame(i)
        assert t == fr.srcTime and good == fr.goodCodes and (codes == fr.codes).all() and (pose == fr.pose).all()
    # an empty frame (no valid depth) is never added
    img, v, n, T = textures(fusion, synth, 0)
    assert not g.addFrame(img, np.zeros_like(v), n, T, 99, 0.05) and not o.addFrame(img, np.zeros_like(v), n, T, 99, 0.05)
    g.close()


def test_find_frame_matches_oracle(mods):
    """findFrame: candidate id, dissimilarity and code agreement exact; where the thumbnail-sized ICP is well
    conditioned (the intra-map query: 10 iterations at one level) the verified pose within the tracker's bar, the accept
    decision, the photometric error and the surface constraints as the restatement gives them.  The 3 x 50 iterations
    of the inter-map query slide along the walls of the synthetic room and end without correspondences on both sides
    (chaotic: not comparable step by step) — both must reject."""
    ferns, fusion, synth, orc, orc_ferns = mods
    g, o = make_pair(mods, photoThresh=1000.0)  # (the colour check would reject the ICP-only pose of this scene)
    for k in range(0, 40, 4):
        img, v, n, T = textures(fusion, synth, k, noise=False)
        assert g.addFrame(img, v, n, T, k, 0.02) == o.addFrame(img, v, n, T, k, 0.02)
    assert len(g) >= 4
    accepted = 0
    for k, interMap, time in ((2, True, 50), (17, True, 60), (30, False, 500), (9, False, 400), (30, False, 100), (250, True, 70)):
        img, v, n, T = textures(fusion, synth, k, noise=False)
        m, cons = g.findFrame(T, v, n, img, time, interMap=interMap)
        r = o.findFrame(T, v, n, img, time, interMap=interMap)
        assert m.candidate == r["candidate"], (k, m.candidate, r["candidate"])
        if r["candidate"] == -1:
            assert m.closest == -1
            continue
        assert np.float32(m.dissimilarity) == np.float32(r["dissimilarity"])
        assert np.float32(m.blockHDAware) == np.float32(r["blockHDAware"])
        if not r["blockHDAware"] > 0.3:
            assert m.closest == -1 and (np.array(m.estPose).reshape(4, 4) == np.eye(4)).all()
            continue
        stable = np.isfinite(r["icp_error"]) and r["icp_count"] > 400
        if not stable:
            assert m.closest == -1 and r["closest"] == -1
            continue
        est = np.array(m.estPose, np.float32).reshape(4, 4)
        helpers.assert_pose_close(est[:3, 3], est[:3, :3], r["estPose"][:3, 3], r["estPose"][:3, :3], what="fern frame %d" % k)
        assert abs(m.icp_count - r["icp_count"]) <= max(5.0, 2e-3 * r["icp_count"])
        assert abs(m.icp_error - r["icp_error"]) <= 2e-3 * max(r["icp_error"], 1e-6)
        assert abs(m.photo_error - r["photo_error"]) <= 0.02 * max(1.0, r["photo_error"])  # a rounding-flipped sample moves it slightly
        assert m.closest == r["closest"]
        if m.closest != -1:
            accepted += 1
            assert m.n_constraints == len(r["constraints"]) and m.n_constraints > 10
            assert np.abs(cons - r["constraints"]).max() < 2e-3  # the poses differ by < 1 mm
    assert accepted >= 1, "at least one revisit must be recognised and verified"
    # the intra-map query only considers frames older than 300 ticks (Ferns.cpp:333)
    img, v, n, T = textures(fusion, synth, 30, noise=False)
    m, _ = g.findFrame(T, v, n, img, 100, interMap=False)
    assert m.candidate == -1
    g.close()


def test_thumbnail_query_equals_texture_query(mods):
    """Collaborative mode hands a remote camera's frame over as the thumbnail block of dms_fusion_thumbnails: the query
    on the block must give what the query on the full-resolution textures gives; search_codes gives the same candidate."""
    ferns, fusion, synth, orc, orc_ferns = mods
    from densemonoslam_amd.capi import DeviceBuffer

    g, o = make_pair(mods)
    for k in range(0, 40, 4):
        img, v, n, T = textures(fusion, synth, k)
        g.addFrame(img, v, n, T, k, 0.02)
    img, v, n, T = textures(fusion, synth, 18)
    m1, c1 = g.findFrame(T, v, n, img, 577, interMap=False)
    ti, tv, tn = o._thumbs(img, v, n)
    block = DeviceBuffer(ti.nbytes + tv.nbytes + tn.nbytes)
    block.upload(np.concatenate([ti.reshape(-1).view(np.uint8), tv.reshape(-1).view(np.uint8), tn.reshape(-1).view(np.uint8)]))
    m2, c2 = g.findFrameThumbs(block.ptr, T, 577, interMap=False)
    assert m1.candidate >= 0 and m1.blockHDAware > 0.3 and m1.icp_count > 100
    for f in ("closest", "candidate", "dissimilarity", "blockHDAware", "icp_error", "icp_count", "photo_error", "n_constraints"):
        assert np.array(getattr(m1, f)).tobytes() == np.array(getattr(m2, f)).tobytes(), f
    assert bytes(m1.estPose) == bytes(m2.estPose) and c1.tobytes() == c2.tobytes()
    # descriptor-only search (what travels between GPUs): same candidate and dissimilarity
    codes, good = g.encode(img, v, n)
    dc, dg, best = DeviceBuffer(512), DeviceBuffer(4), DeviceBuffer(8)
    full = np.full(512, 255, np.uint8)
    full[:500] = codes
    dc.upload(full)
    dg.upload(np.array([good], np.int32))
    g.searchCodes(dc.ptr, dg.ptr, 577, False, best.ptr)
    b = best.download(np.int32, (2,))
    assert b[0] == m1.candidate and np.array([b[1]], np.int32).view(np.float32)[0] == np.float32(m1.dissimilarity)
    g.close()


def test_consume_reencodes_with_the_consumers_table(mods):
    """Ferns::consume (map merge): the other database's frames, re-posed, go through addFrame of the consuming database —
    re-encoded with ITS table (different seed here), accepted or rejected against its frames."""
    ferns, fusion, synth, orc, orc_ferns = mods
    ga, oa = make_pair(mods, seed=3)
    gb, ob = make_pair(mods, seed=4)
    for k in range(0, 24, 4):
        img, v, n, T = textures(fusion, synth, k)
        assert ga.addFrame(img, v, n, T, k, 0.02) == oa.addFrame(img, v, n, T, k, 0.02)
    for k in range(12, 48, 4):
        img, v, n, T = textures(fusion, synth, k)
        assert gb.addFrame(img, v, n, T, 100 + k, 0.02) == ob.addFrame(img, v, n, T, 100 + k, 0.02)
    rel = np.eye(4, dtype=np.float32)
    rel[:3, 3] = (0.5, -0.25, 1.0)
    na, no = ga.consume(gb, rel, 0.02), oa.consume(ob, rel, 0.02)
    assert na == no and 0 < na <= len(ob.frames)
    assert len(ga) == len(oa.frames)
    for i, fr in enumerate(oa.frames):
        pose, t, good, codes = ga.frame(i)
        assert t == fr.srcTime and good == fr.goodCodes and (codes == fr.codes).all() and (pose == fr.pose).all(), i
    ga.close()
    gb.close()


def test_async_add_and_descriptor_search_match_the_synchronous_calls(mods):
    """What collab.InterMapMatcher enqueues per frame — thumbnails of the frame step's fill-in textures, descriptor
    (dms_ferns_encode_thumbs), addFrame without a host synchronisation with the pose read from HBM, descriptor search
    of another camera's block — against the synchronous calls on the same inputs: same database, same candidates."""
    ferns, fusion, synth, orc, orc_ferns = mods
    from densemonoslam_amd.capi import DeviceBuffer
    from densemonoslam_amd import collab

    ef = fusion.ElasticFusion(W, H, K, model_capacity=2_000_000)
    ga, gs = ferns.Ferns(W, H, K, seed=5), ferns.Ferns(W, H, K, seed=5)
    T = collab.thumbnail_bytes(W, H)
    blocks = []
    for k in range(8):
        d, rgb, _ = synth.frame(3 * k, width=W, height=H, K=K, noise=True)
        r = ef.processFrame(rgb, d)
        blk = DeviceBuffer(T + collab.DESC_BYTES)
        if k % 2:  # the one-launch form: thumbnails + pose + tick (what the matcher uses), pose taken from the block's copy
            ef.frameBlock(blk.ptr, blk.ptr + T + collab.DESC_POSE, blk.ptr + T + collab.DESC_TICK, k + 1)
            ref = DeviceBuffer(T)
            ef.thumbnails(ref.ptr)
            whole = blk.download(np.uint8, (T + collab.DESC_BYTES,))
            assert (whole[:T] == ref.download(np.uint8, (T,))).all()
            assert whole[T + collab.DESC_TICK:T + collab.DESC_TICK + 4].view(np.int32)[0] == k + 1
            assert (whole[T + collab.DESC_POSE:T + collab.DESC_POSE + 64].view(np.float32) == np.array(r.pose, np.float32).reshape(16)).all()
            pose_ptr = blk.ptr + T + collab.DESC_POSE
        else:
            ef.thumbnails(blk.ptr)
            pose_ptr = ef.poseDevice()
        if k % 4 == 1:  # descriptor + insertion in one call (one encoding pass, block read in place)
            ga.publishBlock(blk.ptr, blk.ptr + T + collab.DESC_CODES, blk.ptr + T + collab.DESC_GOOD, pose_ptr, k + 1, 0.1)
        else:
            ga.encodeThumbs(blk.ptr, blk.ptr + T + collab.DESC_CODES, blk.ptr + T + collab.DESC_GOOD)
            ga.addFrameAsync(blk.ptr, pose_ptr, k + 1, 0.1)
        if k % 2:
            cd = blk.download(np.uint8, (T + collab.DESC_BYTES,))
            chk = DeviceBuffer(512 + 8)
            ga.encodeThumbs(blk.ptr, chk.ptr, chk.ptr + 512)
            ck = chk.download(np.uint8, (520,))
            assert (cd[T + collab.DESC_CODES:T + collab.DESC_CODES + 512] == ck[:512]).all() and (cd[T + collab.DESC_GOOD:T + collab.DESC_GOOD + 4] == ck[512:516]).all()
        # the synchronous reference call on the fill-in textures themselves
        img, vert, nrm = ef.image(13), ef.image(14), ef.image(15)
        gs.addFrame(img, vert, nrm, np.array(r.pose, np.float32), k + 1, 0.1)
        blocks.append(blk)
    assert len(ga) == len(gs) >= 3
    for i in range(len(gs)):
        pa, ta, ga_, ca = ga.frame(i)
        ps, ts, gs_, cs = gs.frame(i)
        assert ta == ts and ga_ == gs_ and (ca == cs).all() and (pa == ps).all(), i
    # descriptor search of a block == candidate of the full synchronous query on the same block
    best = DeviceBuffer(8)
    for blk in blocks[::3]:
        ga.searchCodes(blk.ptr + T + collab.DESC_CODES, blk.ptr + T + collab.DESC_GOOD, 1000, True, best.ptr)
        b = best.download(np.int32, (2,))
        m, _ = gs.findFrameThumbs(blk.ptr, np.eye(4, dtype=np.float32), 1000, interMap=True)
        assert b[0] == m.candidate and np.array([b[1]], np.int32).view(np.float32)[0] == np.float32(m.dissimilarity)
    # the batched form (all gathered blocks in one launch) gives what the single searches give; the caller's own block is skipped
    nb = len(blocks)
    stride = T + collab.DESC_BYTES
    allb = DeviceBuffer(nb * stride)
    from densemonoslam_amd.capi import lib
    for i, blk in enumerate(blocks):
        assert lib.dms_memcpy_d2d_async(allb.ptr + i * stride, blk.ptr, stride, None) == 0
    bestn = DeviceBuffer(8 * nb)
    ga.searchBlocks(allb.ptr, stride, nb, 2, T + collab.DESC_CODES, T + collab.DESC_GOOD, 1000, True, bestn.ptr)
    bn = bestn.download(np.int32, (nb, 2))
    for i, blk in enumerate(blocks):
        ga.searchCodes(blk.ptr + T + collab.DESC_CODES, blk.ptr + T + collab.DESC_GOOD, 1000, True, best.ptr)
        b = best.download(np.int32, (2,))
        assert (bn[i] == ([-1, -1] if i == 2 else b)).all(), i
    # a second call hands the first call's results out (previous_out) before re-arming the result words
    prev = DeviceBuffer(8 * nb)
    ga.searchBlocks(allb.ptr, stride, nb, -1, T + collab.DESC_CODES, T + collab.DESC_GOOD, 1000, True, bestn.ptr, previous_out=prev.ptr)
    assert (prev.download(np.int32, (nb, 2)) == bn).all()
    bn2 = bestn.download(np.int32, (nb, 2))
    assert (np.delete(bn2, 2, 0) == np.delete(bn, 2, 0)).all() and bn2[2, 0] >= 0  # (nothing skipped this time)
    # a pipelined sequence: from the second call on the search kernel itself hands the previous call's results over and re-arms them
    # (two alternating word sets, no launch between two searches); every call's mirror must hold the call before it
    expect = bn2
    chk = DeviceBuffer(8 * nb)
    for skip in (0, 3, -1, 1, 2):
        ga.searchBlocks(allb.ptr, stride, nb, skip, T + collab.DESC_CODES, T + collab.DESC_GOOD, 1000, True, bestn.ptr, previous_out=prev.ptr)
        assert (prev.download(np.int32, (nb, 2)) == expect).all(), skip
        ga.searchBlocks(allb.ptr, stride, nb, skip, T + collab.DESC_CODES, T + collab.DESC_GOOD, 1000, True, chk.ptr)  # the plain form of the same search
        expect = chk.download(np.int32, (nb, 2))
    # dms_ferns_search_blocks_hd (the pipelined session's per-tick query half): the search's result and, against the frame it chose, the
    # operands of blockHDAware - as the synchronous query reports them; twice (the second kernel re-arms the handle's result words)
    hits = DeviceBuffer(16 * nb)
    for _ in range(2):
        ga.searchBlocksHd(allb.ptr, stride, nb, T + collab.DESC_CODES, T + collab.DESC_GOOD, 1000, True, hits.ptr)
        rows = hits.download(np.int32, (nb, 4))
        for i, blk in enumerate(blocks):
            m, _ = gs.findFrameThumbs(blk.ptr, np.eye(4, dtype=np.float32), 1000, interMap=True)
            assert rows[i, 0] == m.candidate >= 0 and rows[i, 1:2].view(np.float32)[0] == np.float32(m.dissimilarity), i
            assert rows[i, 2] > 0 and np.float32(rows[i, 3]) / np.float32(rows[i, 2]) == np.float32(m.blockHDAware), (i, rows[i], m.blockHDAware)
    ga.close()
    gs.close()
    ef.close()


def test_publish_block_in_one_launch_equals_the_four_step_form(mods, monkeypatch):
    """dms_ferns_publish_block encodes, searches, decides and commits in one launch while the database is small
    (DMS_FERNS_PUBLISH_FUSED, read at creation); the descriptor written into the block, every add / reject decision, the
    stored frames and the count of key frames lost to a full database must equal the four-launch form."""
    ferns, fusion, synth, orc, orc_ferns = mods
    from densemonoslam_amd import collab
    from densemonoslam_amd.capi import DeviceBuffer

    T = collab.thumbnail_bytes(W, H)
    ef = fusion.ElasticFusion(W, H, K, model_capacity=2_000_000)
    blocks = []
    for k in range(10):
        d, rgb, _ = synth.frame(4 * k, width=W, height=H, K=K, noise=True)
        ef.processFrame(rgb, d)
        blk = DeviceBuffer(T + collab.DESC_BYTES)
        ef.frameBlock(blk.ptr, blk.ptr + T + collab.DESC_POSE, blk.ptr + T + collab.DESC_TICK, k + 1)
        blocks.append(blk)
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("DMS_FERNS_PUBLISH_FUSED", fused)
        g = ferns.Ferns(W, H, K, seed=5, capacity=4)  # small: the last key frames find the database full
        descs = []
        for k, blk in enumerate(blocks):
            g.publishBlock(blk.ptr, blk.ptr + T + collab.DESC_CODES, blk.ptr + T + collab.DESC_GOOD, blk.ptr + T + collab.DESC_POSE, k + 1, 0.05)
            descs.append(blk.download(np.uint8, (T + collab.DESC_BYTES,))[T:].copy())
        n = len(g)
        frames = [g.frame(i) for i in range(n)]
        # the batched search + blockHDAware operands of every block against this database: one launch (fused) or two - the same rows
        allb = DeviceBuffer(len(blocks) * (T + collab.DESC_BYTES))
        from densemonoslam_amd.capi import lib

        for i, blk in enumerate(blocks):
            assert lib.dms_memcpy_d2d_async(allb.ptr + i * (T + collab.DESC_BYTES), blk.ptr, T + collab.DESC_BYTES, None) == 0
        hits = DeviceBuffer(16 * len(blocks))
        g.searchBlocksHd(allb.ptr, T + collab.DESC_BYTES, len(blocks), T + collab.DESC_CODES, T + collab.DESC_GOOD, 1000, True, hits.ptr)
        rows = hits.download(np.int32, (len(blocks), 4)).copy()
        out.append((descs, n, frames, g.status() if hasattr(g, "status") else None, rows))
    (da, na, fa, sa, ra), (db, nb, fb, sb, rb) = out
    assert na == nb == 4 and sa == sb
    assert (ra == rb).all() and (ra[:, 0] >= 0).all() and (ra[:, 2] > 0).all(), (ra, rb)
    for x, y in zip(da, db):
        assert (x == y).all()
    for (pa, ta, ga_, ca), (pb, tb, gb_, cb) in zip(fa, fb):
        assert ta == tb and ga_ == gb_ and (ca == cb).all() and (pa == pb).all()


def test_code_agreement_of_exactly_0_3f_verifies_and_wakes(mods):
    """equal / valid = 150 / 500 rounds to 0.3f, which Ferns.cpp:346 compares with the DOUBLE literal 0.3: the candidate is verified.
    find_common (the synchronous query) and the hit row of dms_ferns_search_blocks_hd (the pipelined session's wake rule, formed on the
    host from {valid, equal}) must take that boundary like the reference and the oracle; 149 / 500 stays below on every reading."""
    ferns, fusion, synth, orc, orc_ferns = mods
    from densemonoslam_amd.capi import DeviceBuffer

    up = lambda t: np.ascontiguousarray(np.repeat(np.repeat(t, 8, axis=0), 8, axis=1))  # NEAREST at ratio 8 reads texel 8 j + 4
    I = np.eye(4, dtype=np.float32)
    for equal, verifies in ((150, True), (149, False)):
        g, o = make_pair(mods)
        (ia, va, na), (ib, vb, nb) = helpers.fern_boundary_thumbnails(o, equal)
        assert g.addFrame(up(ia), up(va), up(na), I, 1, 0.3) and o._add(ia, va, na, I, 1, 0.3)
        m, _ = g.findFrame(I, up(vb), up(nb), up(ib), 5, interMap=True)
        r = o.findFrame(I, None, None, None, 5, interMap=True, thumbs=(ib, vb, nb))
        assert m.candidate == r["candidate"] == 0
        assert np.float32(m.blockHDAware).tobytes() == np.float32(r["blockHDAware"]).tobytes() == (np.float32(equal) / np.float32(500)).tobytes()
        assert (m.icp_count > 0) == (r["icp_count"] > 0) == verifies, (equal, m.icp_count, r["icp_count"])
        assert m.icp_count == r["icp_count"] and m.closest == r["closest"]
        # the pipelined session's operands: {candidate, dissimilarity bits, valid in both, equal}
        T = ia.nbytes + va.nbytes + na.nbytes
        blk, rows = DeviceBuffer(T + 1024), DeviceBuffer(16)
        blk.upload(np.concatenate([ib.reshape(-1).view(np.uint8), vb.reshape(-1).view(np.uint8), nb.reshape(-1).view(np.uint8), np.zeros(1024, np.uint8)]))
        g.encodeThumbs(blk.ptr, blk.ptr + T, blk.ptr + T + 512)
        g.searchBlocksHd(blk.ptr, T + 1024, 1, T, T + 512, 0, True, rows.ptr)
        row = rows.download(np.int32, (4,))
        assert list(row[[0, 2, 3]]) == [0, 500, equal]
        assert (float(np.float32(row[3]) / np.float32(row[2])) > 0.3) == verifies == o.searchHit((ib, vb, nb), 5, True)
        g.close()
