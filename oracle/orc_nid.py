"""ORACLE (test infrastructure only) — numpy restatement of the reference's NID score
(computeNIDImg / computeNIDDepth, Cuda/cudafuncs.cu:1513-1650, 1794-1916; histogram kernels
:1086-1157; bin rules :906-918).  PINNED to the reference's own cudafuncs.cu built for gfx950 and run on an MI355X
(oracle/ref_build.sh -> tests/golden/ref_cudafuncs.npz `nid`, held bit for bit by tests/test_ref_cf_pin_cpu.py).

Defined where the reference is not (same rule as the HIP kernels): bins clamped to
[0, num_bins-1], NaN depth -> bin 0."""
import numpy as np


def _pick(d, dold):
    """The reference's validity tests compare with a NaN constant (`dmap != __int_as_float(0x7fffffff)`, :1099-1100), which
    is true for every value, so its first branch always runs: the active view where d <= dold, the old view otherwise (a NaN
    on either side makes the comparison false)."""
    with np.errstate(invalid="ignore"):
        use_new = d <= dold
    return use_new, ~use_new


def _score(hist, num_points):
    """The host half (histogram -> score), in the reference's float / double mix and loop order: oracle/orc_nid.c."""
    import ctypes as C

    from .orc import lib

    hist = np.ascontiguousarray(hist, np.uint32)
    out = np.zeros(4, np.float32)
    lib.orc_nid_score(hist.ctypes.data_as(C.c_void_p), C.c_int(hist.shape[0]), C.c_int(int(num_points)), out.ctypes.data_as(C.c_void_p))
    return np.float32(out[0]), hist


def nid_img(img_kf, img_kf_old, dmap_kf, dmap_kf_old, img_curr, num_bins=64):
    use_new, use_old = _pick(dmap_kf, dmap_kf_old)
    a = np.where(use_new, img_kf, img_kf_old).astype(np.int64)
    b_w = max(256 // num_bins, 1)
    bin_a = np.minimum(a // b_w, num_bins - 1)
    bin_b = np.minimum(img_curr.astype(np.int64) // b_w, num_bins - 1)
    hist = np.zeros((num_bins, num_bins), np.uint32)
    np.add.at(hist, (bin_b.ravel(), bin_a.ravel()), 1)  # row = live, column = key frame (:1117-1118)
    return _score(hist, img_kf.size)


def _bin_depth(mm, max_depth, num_bins):
    b_w = max(int(np.float32(max_depth) / np.float32(num_bins)), 1)
    q = (mm.astype(np.float32) / np.float32(b_w)).astype(np.float32)
    b = np.where(np.isnan(q), 0, np.trunc(np.nan_to_num(q, nan=0.0, posinf=2e9, neginf=-2e9))).astype(np.int64)
    return np.clip(b, 0, num_bins - 1)


def nid_depth(dmap_kf, dmap_kf_old, dmap_curr, num_bins=500, max_depth=25000.0):
    use_new, use_old = _pick(dmap_kf, dmap_kf_old)
    f1000 = np.float32(1000.0)
    a = (np.where(use_new, dmap_kf, dmap_kf_old).astype(np.float32) * f1000).astype(np.float32)  # NaN -> bin 0
    b = (dmap_curr.astype(np.float32) * f1000).astype(np.float32)
    hist = np.zeros((num_bins, num_bins), np.uint32)
    np.add.at(hist, (_bin_depth(b, max_depth, num_bins).ravel(), _bin_depth(a, max_depth, num_bins).ravel()), 1)
    return _score(hist, dmap_kf.size)
