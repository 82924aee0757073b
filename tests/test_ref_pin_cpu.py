"""The CPU restatement of the tracking steps (oracle/orc_track.c) held to the REFERENCE's own kernels.

tests/golden/ref_reduce.npz holds what elasticfusion/Core/src/Cuda/reduce.cu itself returned on an MI355X (built from the
reference's sources by oracle/ref_build.sh, recorded by tests/golden/make_ref_reduce_golden.py) for the cases of
tests/ref_cases.py.  SURVEY 8 rows a2 (icpStep), a3 (computeRgbResidual), a4 (rgbStep), a5 (so3Step), a15 (types)."""
import os

import numpy as np
import pytest

from tests import ref_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_reduce.npz")
# whole-image float sums: the reference adds fp32 partials in its launch order, the restatement in fp64; observed <= 1.5e-6
SUM_RTOL = 1e-5


@pytest.fixture(scope="module")
def golden():
    z = np.load(GOLDEN)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def ours(orc, gputest_pair, golden):
    assert orc.lib.orc_get_fused_rows() == 0  # the source's arithmetic, operation by operation (no contraction)
    lv = ref_cases.inputs(orc, gputest_pair)
    for k, v in ref_cases.input_hashes(lv).items():
        assert str(v) == str(golden[k]), "the restatement is not being fed what the reference saw: " + k
    return ref_cases.run(orc, lv, rows_from=golden)


def _sum_close(a, b, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= SUM_RTOL, "%s: max |d| / max |ref| = %.3e" % (what, err)


def test_fixture_is_the_references_output(golden):
    assert "reduce.cu" in str(golden["meta"]) and "MI355X" in str(golden["meta"])
    assert golden["icp_rows"].shape == (ref_cases.N_ICP_ROWS, 29)
    assert golden["rgb_rows_S0"].shape == (ref_cases.N_RGB_ROWS, 27)


def test_single_pixel_rows_bit_exact(ours, golden):
    """One pixel alone in the image: the reference's sums are that pixel's 27 products, residual and inlier flag."""
    for k in ("icp_rows", "rgb_rows_S0", "rgb_rows_S1", "so3_rows"):
        assert ours[k].dtype == golden[k].dtype == np.float32
        assert np.array_equal(ours[k].view(np.uint32), golden[k].view(np.uint32)), k
    assert golden["icp_rows"][:, 28].sum() >= 0.5 * ref_cases.N_ICP_ROWS  # most chosen pixels do find a correspondence
    assert (np.abs(golden["rgb_rows_S0"]).sum(1) > 0).all()
    assert (golden["so3_rows"][:, 10] == 1).all() and golden["so3_rows"].shape == (ref_cases.N_SO3_ROWS, 11)  # 3 x 3 patches: one pixel each


def test_photometric_correspondences_exact(ours, golden):
    """Every DataTerm field of every pixel and both integer sums, three levels, three relative motions."""
    n = 0
    for lvl in range(3):
        for p in range(len(ref_cases.POSES)):
            key = "rgbres_L%d_P%d" % (lvl, p)
            assert (ours[key + "_sums"] == golden[key + "_sums"]).all(), key
            assert (ours[key + "_valid"] == golden[key + "_valid"]).all(), key
            assert str(ours[key + "_sha"]) == str(golden[key + "_sha"]), key
            n += int(golden[key + "_sums"][1])
    for p in range(len(ref_cases.POSES)):
        a, b = ours["rgbres_L2_P%d_corres" % p], golden["rgbres_L2_P%d_corres" % p]
        for f in a.dtype.names:
            assert (a[f] == b[f]).all(), (p, f)
    assert n > 100000


def test_whole_image_sums(ours, golden):
    for k in golden:
        if k.startswith("icp_L"):
            assert ours[k][28] == golden[k][28] > 1000, "inlier count " + k
            _sum_close(ours[k][:28], golden[k][:28], k)
        elif k.startswith("rgb_L"):
            _sum_close(ours[k], golden[k], k)
        elif k.startswith("so3_L"):
            assert ours[k][10] == golden[k][10] > 1000, "so3 count " + k
            _sum_close(ours[k][:10], golden[k][:10], k)


def test_tracker_loop_around_reference_kernels(orc, gputest_pair, golden):
    """Whole tracker calls (a1): `trk_*` of the fixture were produced by the restatement's host loop (oracle/orc_odometry.c, sums
    in plain order, rows without contraction) calling the REFERENCE's kernels for every step of every iteration on the MI355X
    (orc_odometry_set_step_hooks -> oracle/ref_reduce_harness.cpp ref_hook_*).  The same loop around the restated steps must take
    the same decisions and land on the same pose to within what the different summation orders can move it."""
    from tests import helpers

    ours = ref_cases.run_trackers(orc, gputest_pair)
    for name in ref_cases.TRACKER_CONFIGS:
        k = "trk_%s_" % name
        assert (ours[k + "iters"] == golden[k + "iters"]).all(), (name, ours[k + "iters"], golden[k + "iters"])
        dt = float(np.linalg.norm(ours[k + "t"].astype(np.float64) - golden[k + "t"]))
        da = helpers.rot_angle_deg(ours[k + "R"], golden[k + "R"])
        assert dt <= 1e-4 and da <= 2e-3, (name, dt, da)  # observed <= 1.1e-5 m, 4.2e-4 deg; the bar is 1e-3 m, 1e-2 deg
        assert ours[k + "trace"].shape == golden[k + "trace"].shape
        assert np.abs(ours[k + "trace"] - golden[k + "trace"]).max() <= 2e-4, name  # the pose after EVERY iteration
        c_o, c_g = ours[k + "counts"], golden[k + "counts"]
        assert (np.abs(c_o - c_g) <= 1e-3 * np.maximum(c_g, 1)).all(), (name, c_o, c_g)


def test_fused_mode_against_the_reference_with_default_contraction(orc, gputest_pair):
    """The resident tracker evaluates its rows with every multiply-add chain fused (what nvcc's default -fmad=true is to the
    reference's own build); the restatement's FUSED mode is its checker.  tests/golden/ref_reduce_fma.npz holds the reference's
    kernels built with the COMPILER's default contraction (oracle/ref_build.sh, second library).  A compiler picks its own
    fusions inside a cross product, so rows agree to the last bits rather than bit for bit (observed: 53-73 % of the rows
    identical, the rest within 2.9e-7 of the row's largest entry) — but every DECISION is the same: all ~170 000 photometric
    correspondences with their fields, the inlier counts, the iteration counts of whole tracker calls."""
    from tests import helpers

    z = np.load(os.path.join(os.path.dirname(GOLDEN), "ref_reduce_fma.npz"))
    gold = {k: z[k] for k in z.files}
    base = np.load(GOLDEN)
    lv = ref_cases.inputs(orc, gputest_pair)
    for k, v in ref_cases.input_hashes(lv).items():
        assert str(v) == str(gold[k]), k
    orc.lib.orc_set_fused_rows(1)
    try:
        ours = ref_cases.run(orc, lv, rows_from={k: base[k] for k in ("icp_row_pix", "rgb_row_pix", "rgbres_L2_P1_corres")})
    finally:
        orc.lib.orc_set_fused_rows(0)
    for lvl in range(3):
        for p in range(len(ref_cases.POSES)):
            key = "rgbres_L%d_P%d" % (lvl, p)
            assert (ours[key + "_sums"] == gold[key + "_sums"]).all(), key
            assert str(ours[key + "_sha"]) == str(gold[key + "_sha"]), key
            assert ours["icp_L%d_P%d" % (lvl, p)][28] == gold["icp_L%d_P%d" % (lvl, p)][28], key
    for k in ("icp_rows", "rgb_rows_S0", "rgb_rows_S1", "so3_rows"):
        a, b = ours[k].astype(np.float64), gold[k].astype(np.float64)
        scale = np.maximum(np.abs(b).max(1, keepdims=True), 1e-30)
        assert (np.abs(a - b) / scale).max() <= 1e-6, k
        assert (ours[k][:, -1] == gold[k][:, -1]).all(), k  # the found / inlier flag of every isolated pixel
        assert (ours[k].view(np.uint32) == gold[k].view(np.uint32)).all(1).mean() >= 0.45, k
    for k in gold:
        if k.startswith(("icp_L", "rgb_L", "so3_L")):
            n = 28 if k.startswith("icp_L") else (10 if k.startswith("so3_L") else 27)
            _sum_close(ours[k][:n], gold[k][:n], k)
    trk = ref_cases.run_trackers(orc, gputest_pair, fused=True)
    for name in ref_cases.TRACKER_CONFIGS:
        k = "trk_%s_" % name
        assert (trk[k + "iters"] == gold[k + "iters"]).all(), name
        dt = float(np.linalg.norm(trk[k + "t"].astype(np.float64) - gold[k + "t"]))
        da = helpers.rot_angle_deg(trk[k + "R"], gold[k + "R"])
        # two different sets of fusions through up to 29 iterations: observed <= 7.1e-5 m, 2.9e-3 deg (the photometric-only
        # configuration, the weakest conditioned); half the north-star bar is required
        assert dt <= 5e-4 and da <= 5e-3, (name, dt, da)
