"""Test infrastructure: ctypes binding of oracle/_ref/libref_reduce.so = the REFERENCE's own tracking steps
(elasticfusion/Core/src/Cuda/reduce.cu) built for gfx950 by oracle/ref_build.sh.  Same call signatures as the
restatement in oracle/orc.py (icpStep, computeRgbResidual, rgbStep, so3Step) so one case runner drives both.

Needs a GPU (the reference's kernels run on it).  Used by tests/golden/make_ref_reduce_golden.py (which records the
reference's outputs as committed fixtures) and by the `-m gpu` test that compares the HIP path with the reference live.
Never imported by the product."""
import ctypes as C
import os

import numpy as np

# DMS_REF_VARIANT=fma selects the build with the compiler's default contraction (oracle/ref_build.sh); read when the library is first used
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref",
                        "libref_reduce_fma.so" if os.environ.get("DMS_REF_VARIANT") == "fma" else "libref_reduce.so")

# DataTerm as the reference lays it out (Cuda/types.cuh:75-81): short2, short2, float, bool (+3 bytes of padding)
REF_DATATERM = np.dtype([("zero_x", "<i2"), ("zero_y", "<i2"), ("one_x", "<i2"), ("one_y", "<i2"), ("diff", "<f4"), ("valid", "u1"),
                         ("pad", "u1", (3,))])
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref_reduce.so is missing: run oracle/ref_build.sh where /root/reference exists")
        _lib = C.CDLL(LIB_PATH)
        sizes = (C.c_int * 4)()
        _lib.ref_sizes(sizes)
        assert list(sizes) == [116, 44, 16, 36], list(sizes)
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(v):
    return C.c_float(float(v))


def icpStep(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, distThres, angleThres,
            threads=128, blocks=112):
    a = [_c(Rcurr, np.float32).reshape(9), _c(tcurr, np.float32).reshape(3), _c(vmap_curr, np.float32), _c(nmap_curr, np.float32),
         _c(Rprev_inv, np.float32).reshape(9), _c(tprev, np.float32).reshape(3), _c(cam, np.float32).reshape(4),
         _c(vmap_g_prev, np.float32), _c(nmap_g_prev, np.float32)]
    rows, cols = a[2].shape[0] // 3, a[2].shape[1]
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    res = np.zeros(2, np.float32)
    rc = lib().ref_icpStep(*[_p(x) for x in a], rows, cols, _f(distThres), _f(angleThres), threads, blocks, _p(A), _p(b), _p(res))
    assert rc == 0, rc
    return A, b, res


def to_orc_dataterm(raw):
    """Reference records -> the oracle's dtype; fields of records without a correspondence are not written by the
    reference's kernel (reduce.cu:776-838 sets only `valid`), so they are zeroed here."""
    from oracle.orc import DATATERM_DTYPE

    out = np.zeros(raw.shape, DATATERM_DTYPE)
    v = raw["valid"] != 0
    for f in ("zero_x", "zero_y", "one_x", "one_y", "diff"):
        out[f] = np.where(v, raw[f], 0)
    out["valid"] = v
    return out


def computeRgbResidual(minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, maxDepthDelta, kt, krkinv, threads=256,
                       blocks=336):
    dIdx, dIdy = _c(dIdx, np.int16), _c(dIdy, np.int16)
    lastDepth, nextDepth = _c(lastDepth, np.float32), _c(nextDepth, np.float32)
    lastImage, nextImage = _c(lastImage, np.uint8), _c(nextImage, np.uint8)
    kt = _c(kt, np.float32).reshape(3)
    krkinv = _c(krkinv, np.float32).reshape(9)
    rows, cols = nextImage.shape
    raw = np.zeros((rows, cols), REF_DATATERM)
    sigma, count = C.c_int(0), C.c_int(0)
    rc = lib().ref_computeRgbResidual(_f(minScale), _p(dIdx), _p(dIdy), _p(lastDepth), _p(nextDepth), _p(lastImage), _p(nextImage),
                                      rows, cols, _f(maxDepthDelta), _p(kt), _p(krkinv), threads, blocks, _p(raw), C.byref(sigma),
                                      C.byref(count))
    assert rc == 0, rc
    return to_orc_dataterm(raw), sigma.value, count.value


def rgbStep(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale, threads=128, blocks=112):
    raw = np.zeros(corres.shape, REF_DATATERM)
    for f in ("zero_x", "zero_y", "one_x", "one_y", "diff"):
        raw[f] = corres[f]
    raw["valid"] = corres["valid"] != 0
    cloud = _c(cloud, np.float32)
    dIdx, dIdy = _c(dIdx, np.int16), _c(dIdy, np.int16)
    rows, cols = dIdx.shape
    A = np.zeros((6, 6), np.float32)
    b = np.zeros(6, np.float32)
    rc = lib().ref_rgbStep(_p(raw), _f(sigma), _p(cloud), _f(fx), _f(fy), _p(dIdx), _p(dIdy), _f(sobelScale), rows, cols, threads,
                           blocks, _p(A), _p(b))
    assert rc == 0, rc
    return A, b


def so3Step(lastImage, nextImage, imageBasis, kinv, krlr, threads=128, blocks=64):
    lastImage, nextImage = _c(lastImage, np.uint8), _c(nextImage, np.uint8)
    ib, ki, kr = (_c(m, np.float32).reshape(9) for m in (imageBasis, kinv, krlr))
    rows, cols = nextImage.shape
    A = np.zeros((3, 3), np.float32)
    b = np.zeros(3, np.float32)
    res = np.zeros(2, np.float32)
    rc = lib().ref_so3Step(_p(lastImage), _p(nextImage), _p(ib), _p(ki), _p(kr), rows, cols, threads, blocks, _p(A), _p(b), _p(res))
    assert rc == 0, rc
    return A, b, res


def step_hooks():
    """Addresses of the reference's four steps behind the restatement's signatures (oracle/ref_reduce_harness.cpp ref_hook_*),
    in the order of orc.Odometry.setStepHooks."""
    L = lib()
    return tuple(C.cast(getattr(L, n), C.c_void_p).value for n in ("ref_hook_so3Step", "ref_hook_computeRgbResidual", "ref_hook_icpStep",
                                                                   "ref_hook_rgbStep"))
