"""The HIP path's tracking steps (through the C ABI) against what the REFERENCE's own kernels returned
(tests/golden/ref_reduce.npz, see tests/test_ref_pin_cpu.py): not product-vs-restatement but product-vs-reference."""
import os

import numpy as np
import pytest

from tests import ref_cases

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_reduce.npz")
SUM_RTOL = 2e-6  # sums in two different orders (the product's tree, the reference's strided fp32 threads); observed <= 1.1e-7


class _Ops:
    """dms.ops with computeRgbResidual's device image brought to the host (the runner hashes it)."""

    def __init__(self, ops):
        self.ops = ops
        self.icpStep, self.rgbStep, self.so3Step = ops.icpStep, ops.rgbStep, ops.so3Step

    def computeRgbResidual(self, *a):
        c, s, n = self.ops.computeRgbResidual(*a)
        return c.download(), s, n


@pytest.fixture(scope="module")
def pair(orc, gputest_pair):
    from densemonoslam_amd import capi, odometry

    assert capi.device_count() >= 1, "no MI355X visible"
    z = np.load(GOLDEN)
    golden = {k: z[k] for k in z.files}
    lv = ref_cases.inputs(orc, gputest_pair)  # inputs only (hash-checked against what the reference saw)
    for k, v in ref_cases.input_hashes(lv).items():
        assert str(v) == str(golden[k]), k
    return ref_cases.run(_Ops(odometry.ops), lv, rows_from=golden), golden


def test_product_rows_and_correspondences_equal_the_references(pair):
    ours, golden = pair
    for k in ("icp_rows", "rgb_rows_S0", "rgb_rows_S1", "so3_rows"):
        # one pixel alone in the image: its 27 products, residual and inlier flag, bit for bit what the reference's kernel formed
        assert np.array_equal(ours[k].view(np.uint32), golden[k].view(np.uint32)), k
    for lvl in range(3):
        for p in range(len(ref_cases.POSES)):
            key = "rgbres_L%d_P%d" % (lvl, p)
            assert (ours[key + "_sums"] == golden[key + "_sums"]).all(), key
            assert (ours[key + "_valid"] == golden[key + "_valid"]).all(), key
            assert str(ours[key + "_sha"]) == str(golden[key + "_sha"]), key


def test_product_sums_equal_the_references(pair):
    ours, golden = pair

    def close(a, b, what):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err <= SUM_RTOL, "%s: %.3e" % (what, err)

    for k in golden:
        if k.startswith("icp_L"):
            assert ours[k][28] == golden[k][28], k
            close(ours[k][:28], golden[k][:28], k)
        elif k.startswith("rgb_L"):
            close(ours[k], golden[k], k)
        elif k.startswith("so3_L"):
            assert ours[k][10] == golden[k][10], k
            close(ours[k][:10], golden[k][:10], k)


@pytest.mark.parametrize("fixture", ["ref_reduce.npz", "ref_reduce_fma.npz"])
def test_product_tracker_against_the_reference_driven_tracker(orc, gputest_pair, fixture):
    """getIncrementalTransformation of the HIP path (resident kernels, canonical sums, fused rows) against tracker calls whose
    every step ran the REFERENCE's kernels (`trk_*` of the fixture, see tests/test_ref_pin_cpu.py): the north-star bar, 1 mm and
    0.01 degree, and the same iteration counts."""
    from densemonoslam_amd import odometry
    from tests import helpers

    # (ref_reduce_fma.npz: the reference's kernels built with the compiler's default contraction, see tests/test_ref_pin_cpu.py)
    z = np.load(os.path.join(os.path.dirname(GOLDEN), fixture))
    K = ref_cases.K
    verts, norms = helpers.gputest_model_maps(gputest_pair["depth1_raw"], K)
    worst = (0.0, 0.0)
    for name, cfg in ref_cases.TRACKER_CONFIGS.items():
        g = odometry.RGBDOdometry(640, 480, K[2], K[3], K[0], K[1])
        g.initICPModel(verts, norms, 20.0, np.eye(4, dtype=np.float32))
        g.initRGBModel(helpers.rgba(gputest_pair["rgb1"]))
        g.initICP(gputest_pair["depth2"], 20.0)
        g.initRGB(helpers.rgba(gputest_pair["rgb2"]))
        g.initFirstRGB(helpers.rgba(gputest_pair["rgb1"]))
        t, R, res = g.getIncrementalTransformation(np.zeros(3, np.float32), np.eye(3, dtype=np.float32), **cfg)
        dt, da = helpers.assert_pose_close(t, R, z["trk_%s_t" % name], z["trk_%s_R" % name], tol_m=1e-3, tol_deg=1e-2, what=name)
        worst = (max(worst[0], dt), max(worst[1], da))
        assert [res.so3_iterations_run] + list(res.iterations_run) == list(z["trk_%s_iters" % name]), name
        g.close()
    print("worst difference to the reference-driven tracker (%s): %.2e m, %.2e deg" % (fixture, worst[0], worst[1]))
