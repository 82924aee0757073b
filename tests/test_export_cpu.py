"""End-of-run exports (host-side code of the library; no GPU needed for the trajectory)."""
import ctypes as C
import os

import numpy as np

from oracle import orc_export


def test_trajectory_file_matches_the_reference_format(tmp_path):
    """Context::saveTrajectory (Context.h:117-156): `f << rot(0,0) << " " << ... << trans(2) << " " << "\\n"` with an
    ostream's default float formatting.  Known lines written out by hand + the restatement on random poses."""
    from densemonoslam_amd import capi

    rng = np.random.default_rng(3)
    poses = np.zeros((5, 4, 4), np.float32)
    poses[:] = np.eye(4)
    poses[1, :3, 3] = [0.5, -1.25, 1e-7]
    poses[2, :3, :] = rng.standard_normal((3, 4)).astype(np.float32)
    poses[3, :3, :] = (rng.standard_normal((3, 4)) * 1e6).astype(np.float32)
    poses[4, :3, :] = (rng.standard_normal((3, 4)) * 1e-6).astype(np.float32)
    path = str(tmp_path / "cam.freiburg")
    rc = capi.lib.dms_trajectory_save(os.fsencode(path), poses.ctypes.data_as(C.c_void_p), C.c_size_t(len(poses)))
    assert rc == 0
    text = open(path).read()
    lines = text.splitlines(keepends=True)
    assert lines[0] == "1 0 0 0 0 1 0 0 0 0 1 0 \n"
    assert lines[1] == "1 0 0 0.5 0 1 0 -1.25 0 0 1 1e-07 \n"
    assert text == orc_export.trajectory_text(poses)
    assert capi.lib.dms_trajectory_save(None, None, C.c_size_t(0)) != 0


def test_ply_restatement_layout():
    """The checker itself: 31 bytes per vertex after the header, confidence gate strict, colour unpacked from the float."""
    rec = np.zeros((3, 15), np.float32)
    rec[:, 3] = [0.5, 2.0, 3.0]
    rec[:, 0:3] = [[1, 2, 3], [4, 5, 6], [7, 8, 9]]
    rec[:, 4] = float((10 << 16) | (20 << 8) | 30)
    rec[:, 11:15] = [[0, 0, 1, 0.1], [0, 1, 0, 0.2], [1, 0, 0, 0.3]]
    b = orc_export.save_ply_bytes(rec, 1.0)
    head, body = b.split(b"end_header\n")
    assert b"element vertex 2\n" in head and len(body) == 2 * 31
    v = np.frombuffer(body[:12], "<f4")
    assert list(v) == [4, 5, 6] and list(body[12:15]) == [10, 20, 30]
    assert list(np.frombuffer(body[15:31], "<f4")) == [0, -1, 0, np.float32(0.2)]
    # the reference's stale + 18 offset lands in the next record (confidence, colour, colour...), zeros past the end
    b2 = orc_export.save_ply_bytes(rec, 1.0, reference_offsets=True).split(b"end_header\n")[1]
    assert list(np.frombuffer(b2[15:31], "<f4"))[0] == -np.float32(3.0)
    assert list(np.frombuffer(b2[31 + 15:62], "<f4")) == [0, 0, 0, 0]
