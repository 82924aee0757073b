"""Test infrastructure: ctypes binding of oracle/_ref/libref_gl.so (oracle/ref_gl_harness.c) = the REFERENCE's own GLSL programs
(elasticfusion/Core/src/Shaders/*.vert, *.geom, *.frag, read from the reference tree at run time) executed by the image's Mesa
llvmpipe, a software OpenGL 4.5.  Same call signatures and the same array conventions as the restatement's fusion half in
oracle/orc.py (depth_bilateral, depth_metric, model_initialise, index_map, splat_predict, model_fuse, model_clean, fill_in) so one
case runner drives both; surfels travel as orc.SURFEL_DTYPE records and are converted to / from the reference's 15-float layout
(Shaders/Vertex.cpp:21-50) here.

Runs on the CPU (no GPU needed) but only where the reference's shader files exist: tests/golden/make_ref_glsl_golden.py uses it in
the build container to record fixtures; the tests then compare the restatement (CPU) and the product (GPU) with those fixtures.
Never imported by the product."""
import ctypes as C
import os

import numpy as np

from .orc import SURFEL_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref_gl.so")
SHADER_DIR = os.path.join(os.environ.get("DMS_REFERENCE_ROOT", "/root/reference"), "elasticfusion", "Core", "src", "Shaders")
TEX_DIM = 5700  # GlobalModel::TEXTURE_DIMENSION (GlobalModel.cpp:22)
_lib = None


def available():
    return os.path.exists(LIB_PATH) and os.path.exists(os.path.join(SHADER_DIR, "index_map.vert"))


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref_gl.so or the reference's Shaders/ directory is missing")
        # the reference's fill_rgb.frag / resize.frag call texture2D under `#version 440 core`; NVIDIA's compiler accepts that, Mesa's
        # needs its documented leniency switch (driconf option, read from the environment)
        os.environ.setdefault("force_compat_shaders", "true")
        L = C.CDLL(LIB_PATH)
        L.rgl_error.restype = C.c_char_p
        L.rgl_renderer.restype = C.c_char_p
        L.rgl_version.restype = C.c_char_p
        if L.rgl_init(SHADER_DIR.encode()) != 0:
            raise RuntimeError("rgl_init: " + L.rgl_error().decode())
        _lib = L
    return _lib


def renderer():
    L = lib()
    return "%s / OpenGL %s" % (L.rgl_renderer().decode(), L.rgl_version().decode())


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f(v):
    return C.c_float(float(v))


def _ok(rc, what):
    if rc < 0:
        raise RuntimeError("%s: %s" % (what, lib().rgl_error().decode()))
    return rc


def to_ref(model):
    """orc.SURFEL_DTYPE records -> n x 15 floats (pos.xyz conf | colour 0 initTime stamp | times[3] | normal.xyz radius)."""
    model = np.ascontiguousarray(model, SURFEL_DTYPE)
    out = np.zeros((len(model), 15), np.float32)
    out[:, 0:4] = model["pos"]
    out[:, 4:8] = model["col"]
    out[:, 8:11] = model["times"][:, :3]
    out[:, 11:15] = model["nrm"]
    return out


def from_ref(a):
    a = np.asarray(a, np.float32).reshape(-1, 15)
    out = np.zeros(len(a), SURFEL_DTYPE)
    out["pos"] = a[:, 0:4]
    out["col"] = a[:, 4:8]
    out["times"][:, :3] = a[:, 8:11]
    out["times"][:, 3:] = -3.0  # the product's / restatement's wider record: sensors the reference does not have have not seen it
    out["nrm"] = a[:, 11:15]
    return out


def inv4(pose):
    """pose.inverse() of the reference's host code (Eigen, absent here): computed in float64 and rounded, which for a rigid
    transform agrees with Eigen's float inverse to the last bits; callers that compare bit for bit hand the SAME t_inv to both
    sides (orc.inv4f restates the float path)."""
    return np.linalg.inv(np.asarray(pose, np.float64).reshape(4, 4)).astype(np.float32)


def depth_bilateral(depth, maxD):
    d = _c(depth, np.uint16)
    out = np.zeros_like(d)
    _ok(lib().rgl_depth_bilateral(_p(d), d.shape[0], d.shape[1], _f(maxD), _p(out)), "depth_bilateral")
    return out


def depth_metric(depth, maxD):
    d = _c(depth, np.uint16)
    out = np.zeros(d.shape, np.float32)
    _ok(lib().rgl_depth_metric(_p(d), d.shape[0], d.shape[1], _f(maxD), _p(out)), "depth_metric")
    return out


def vertex_feedback(rgba, dm, cam, time, timeIdx, maxDepth, depth_linear):
    rgba, dm = _c(rgba, np.uint8), _c(dm, np.float32)
    rows, cols = dm.shape
    out = np.zeros((rows * cols, 15), np.float32)
    n = _ok(lib().rgl_vertex_feedback(_p(rgba), _p(dm), rows, cols, _f(cam[2]), _f(cam[3]), _f(cam[0]), _f(cam[1]), int(time), int(timeIdx),
                                      _f(maxDepth), int(depth_linear), _p(out)), "vertex_feedback")
    return out[:n].copy()


def model_initialise(rgba, dm, dmf, cam, time, timeIdx, maxDepth):
    """FeedbackBuffer::compute on the raw and on the filtered metric depth, then GlobalModel::initialise (ElasticFusion.cpp:117-129)."""
    raw = vertex_feedback(rgba, dm, cam, time, timeIdx, maxDepth, True)
    fil = vertex_feedback(rgba, dmf, cam, time, timeIdx, maxDepth, False)
    assert len(raw) == len(fil), "raw / filtered feedback buffers have different lengths: the reference would pair mismatched streams"
    out = np.zeros((max(len(raw), 1), 15), np.float32)
    n = _ok(lib().rgl_model_initialise(_p(raw), _p(fil), len(raw), _p(out)), "model_initialise")
    return from_ref(out[:n])


def index_map(model, pose, cam, rows, cols, time, timeIdx, maxDepth, timeDelta, t_inv=None):
    m = to_ref(model)
    ti = _c(inv4(pose) if t_inv is None else t_inv, np.float32).reshape(16)
    index = np.zeros((rows, cols), np.uint32)
    vc, ct, nr = (np.zeros((rows, cols, 4), np.float32) for _ in range(3))
    _ok(lib().rgl_index_map(_p(m), len(m), _p(ti), _f(cam[2]), _f(cam[3]), _f(cam[0]), _f(cam[1]), rows, cols, int(time), int(timeIdx),
                            _f(maxDepth), int(timeDelta), _p(index), _p(vc), _p(ct), _p(nr)), "index_map")
    return index, vc, ct, nr


def splat_predict(model, pose, cam, rows, cols, maxDepth, confThreshold, time, timeIdx, maxTime, timeDelta, active, depth_only=False,
                  t_inv=None):
    m = to_ref(model)
    ti = _c(inv4(pose) if t_inv is None else t_inv, np.float32).reshape(16)
    a = [_p(m), len(m), _p(ti), _f(cam[2]), _f(cam[3]), _f(cam[0]), _f(cam[1]), rows, cols, _f(maxDepth), _f(confThreshold), int(time),
         int(timeIdx), int(maxTime), int(timeDelta), int(bool(active))]
    if depth_only:
        d = np.zeros((rows, cols), np.float32)
        _ok(lib().rgl_splat(*a, 1, None, None, None, None, _p(d)), "synthesizeDepth")
        return d
    image = np.zeros((rows, cols, 4), np.uint8)
    vertex = np.zeros((rows, cols, 4), np.float32)
    normal = np.zeros((rows, cols, 4), np.float32)
    timg = np.zeros((rows, cols), np.uint16)
    _ok(lib().rgl_splat(*a, 0, _p(image), _p(vertex), _p(normal), _p(timg), None), "combinedPredict")
    return image, vertex, normal, timg


def model_fuse(model, pose, time, timeIdx, rgba, dr, drf, index, vertConf, colorTime, normRad, cam, maxDepth, weighting, texDim=TEX_DIM):
    """Returns (updated model, every record the data pass emitted into the feedback buffer)."""
    m = to_ref(model)
    pose = _c(pose, np.float32).reshape(16)
    rgba, dr, drf = _c(rgba, np.uint8), _c(dr, np.float32), _c(drf, np.float32)
    index, vertConf, colorTime, normRad = _c(index, np.uint32), _c(vertConf, np.float32), _c(colorTime, np.float32), _c(normRad, np.float32)
    rows, cols = dr.shape
    out = np.zeros((max(len(m), 1), 15), np.float32)
    newU = np.zeros((rows * cols, 15), np.float32)
    n = _ok(lib().rgl_model_fuse(_p(m), len(m), _p(pose), int(time), int(timeIdx), _p(rgba), _p(dr), _p(drf), _p(index), _p(vertConf),
                                 _p(colorTime), _p(normRad), rows, cols, _f(cam[2]), _f(cam[3]), _f(cam[0]), _f(cam[1]), _f(maxDepth),
                                 _f(weighting), int(texDim), _p(out), _p(newU)), "model_fuse")
    return from_ref(out[:len(m)]), from_ref(newU[:n])


def model_clean(model, newU, pose, time, timeIdx, index, vertConf, colorTime, normRad, cam, confThreshold, timeDelta, maxDepth, nodes=None,
                depthSynth=None, isFern=0, t_inv=None):
    m, nu = to_ref(model), to_ref(newU)
    ti = _c(inv4(pose) if t_inv is None else t_inv, np.float32).reshape(16)
    index, vertConf, colorTime, normRad = _c(index, np.uint32), _c(vertConf, np.float32), _c(colorTime, np.float32), _c(normRad, np.float32)
    rows, cols = index.shape
    nn, nptr = 0, None
    if nodes is not None and len(nodes):
        nodes = _c(nodes, np.float32).reshape(-1, 16)
        nn, nptr = len(nodes), _p(nodes)
    if depthSynth is not None:
        depthSynth = _c(depthSynth, np.float32)
    out = np.zeros((len(m) + len(nu) + 1, 15), np.float32)
    n = _ok(lib().rgl_model_clean(_p(m), len(m), _p(nu), len(nu), _p(ti), int(time), int(timeIdx), _p(index), _p(vertConf), _p(colorTime),
                                  _p(normRad), _p(depthSynth), rows, cols, _f(cam[2]), _f(cam[3]), _f(cam[0]), _f(cam[1]), _f(confThreshold),
                                  nptr, nn, int(timeDelta), _f(maxDepth), int(isFern), _p(out)), "model_clean")
    return from_ref(out[:n])


def fill(which, existing, depth, cam, passthrough):
    """FillIn::vertex (which = 0) / FillIn::normal (which = 1)."""
    existing, depth = _c(existing, np.float32), _c(depth, np.uint16)
    rows, cols = depth.shape
    out = np.zeros((rows, cols, 4), np.float32)
    _ok(lib().rgl_fill(int(which), _p(existing), _p(depth), rows, cols, _f(cam[2]), _f(cam[3]), _f(cam[0]), _f(cam[1]), int(bool(passthrough)),
                       _p(out)), "fill")
    return out


def model_consume(dst, src, relativeTransform):
    """GlobalModel::consume: dst's records unchanged, then src's moved by relativeTransform"""
    d, m = to_ref(dst), to_ref(src)
    T = _c(relativeTransform, np.float32).reshape(16)
    out = np.zeros((max(len(d) + len(m), 1), 15), np.float32)
    n = _ok(lib().rgl_model_consume(_p(d), len(d), _p(m), len(m), _p(T), _p(out)), "model_consume")
    return from_ref(out[:n])


def sample_graph(model, sampleRate, timeIdx=0):
    """Deformation::sampleGraphModel: the program's samples (map order) and then the host's sort by init time (stable here)"""
    m = to_ref(model)
    out = np.zeros((max(len(m), 1), 4), np.float32)
    n = _ok(lib().rgl_graph_sample(_p(m), len(m), int(timeIdx), int(sampleRate), _p(out)), "graph_sample")
    rows = out[:n].copy()
    return rows[np.argsort(rows[:, 3], kind="stable")]


def fill_rgb(existing_rgba, raw_rgba, passthrough):
    """FillIn::image"""
    e, r = _c(existing_rgba, np.uint8), _c(raw_rgba, np.uint8)
    out = np.zeros_like(e)
    _ok(lib().rgl_fill_rgb(_p(e), _p(r), e.shape[0], e.shape[1], int(bool(passthrough)), _p(out)), "fill_rgb")
    return out


def resize(src, drows, dcols):
    """Resize::image (uint8 H x W x 4) / Resize::vertex (float32 H x W x 4)"""
    src = np.ascontiguousarray(src)
    which = 1 if src.dtype == np.float32 else 0
    dst = np.zeros((drows, dcols, 4), src.dtype)
    _ok(lib().rgl_resize(which, _p(src), src.shape[0], src.shape[1], int(drows), int(dcols), _p(dst)), "resize")
    return dst


def try_program(vs, gs, fs):
    return lib().rgl_try_program(vs.encode(), (gs or "").encode(), (fs or "").encode()) == 0, lib().rgl_error().decode()
