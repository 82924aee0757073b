"""ORACLE (test infrastructure only) — numpy restatement of the reference's NID score
(computeNIDImg / computeNIDDepth, Cuda/cudafuncs.cu:1513-1650, 1794-1916; histogram kernels
:1086-1157; bin rules :906-918).  PARITY UNPINNED: the reference holds no test vector for it and
cannot be built here; the restatement follows the host loops line by line in float32.

Defined where the reference is not (same rule as the HIP kernels): bins clamped to
[0, num_bins-1], NaN depth -> bin 0."""
import numpy as np


def _pick(d, dold):
    v, vo = ~np.isnan(d), ~np.isnan(dold)
    use_new = v & (~vo | (d <= dold))
    use_old = vo & ~use_new
    return use_new, use_old


def _score(hist, num_points):
    nb = hist.shape[0]
    if num_points == 0:
        return np.float32(1.0), hist
    h = (hist.astype(np.float32) / np.float32(num_points)).astype(np.float32)  # histogram_host[...] /= num_points
    PB = np.zeros(nb, np.float32)
    PA = np.zeros(nb, np.float32)
    for k in range(nb):  # float accumulation in the reference's loop order (:1571-1587)
        PB[k] = np.add.reduce(h[:, k], dtype=np.float32)
        PA[k] = np.add.reduce(h[k, :], dtype=np.float32)

    def plogp(x):
        x = x.astype(np.float32)
        out = np.zeros_like(x)
        nz = x != 0
        out[nz] = x[nz] * np.log2(x[nz]).astype(np.float32)
        return out

    joint = -np.float32(plogp(h).sum(dtype=np.float64))
    kf = -np.float32(plogp(PB).sum(dtype=np.float64))
    cf = -np.float32(plogp(PA).sum(dtype=np.float64))
    mi = np.float32(kf + cf - joint)
    return np.float32((joint - mi) / joint), hist


def nid_img(img_kf, img_kf_old, dmap_kf, dmap_kf_old, img_curr, num_bins=64):
    use_new, use_old = _pick(dmap_kf, dmap_kf_old)
    a = np.where(use_new, img_kf, np.where(use_old, img_kf_old, 0)).astype(np.int64)
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
    a = np.where(use_new, dmap_kf.astype(np.float32) * f1000, np.where(use_old, dmap_kf_old.astype(np.float32) * f1000, np.float32(0))).astype(np.float32)
    b = (dmap_curr.astype(np.float32) * f1000).astype(np.float32)
    hist = np.zeros((num_bins, num_bins), np.uint32)
    np.add.at(hist, (_bin_depth(b, max_depth, num_bins).ravel(), _bin_depth(a, max_depth, num_bins).ravel()), 1)
    return _score(hist, dmap_kf.size)
