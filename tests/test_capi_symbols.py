"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports
every entry point that include/*.h declares; argument validation works without device access."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        names += re.findall(r"\b(dms_[a-zA-Z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from densemonoslam_amd import capi

    names = declared_functions()
    assert len(names) >= 40, names
    missing = [n for n in names if not hasattr(capi.lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_version_and_error_text_without_gpu():
    from densemonoslam_amd import capi

    assert b"gfx950" in capi.lib.dms_version()
    assert capi.lib.dms_reduce_workspace_bytes() >= 29 * 1024 * 4
    # argument validation happens before any device access
    assert capi.lib.dms_createNMap(None, None, None) == -1
    assert b"null" in capi.lib.dms_last_error()
    n = ctypes.c_int(-1)
    rc = capi.lib.dms_device_count(ctypes.byref(n))
    assert rc in (0, -2)


def test_struct_sizes_match_reference_types():
    """JtJJtrSE3 = 29 floats, DataTerm = 16 bytes (Cuda/types.cuh:77-83,123-171)."""
    from densemonoslam_amd import capi

    assert capi.DATATERM_DTYPE.itemsize == 16
    assert ctypes.sizeof(capi.Image2D) == 24
    assert ctypes.sizeof(capi.Mat33) == 36
    assert ctypes.sizeof(capi.Camera) == 16


def test_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: the three public headers compile as C99 (and as C++11) on their own,
    without HIP or torch types."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi_check.c"
    src.write_text('#include "dmslam.h"\n#include "dmslam_fusion.h"\n#include "dmslam_io.h"\n'
                   "int main(void) { dms_fusion_params p; dms_frame_result r; dms_frame_msg m; (void)p; (void)r; (void)m; return 0; }\n")
    inc = os.path.join(root, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])


def test_thumbnail_block_layout_for_any_frame_size():
    """dms_thumb_block_bytes / _offsets (include/dmslam_fusion.h): W/8 x H/8 pixels with truncating division (Ferns.cpp:23-24), the RGBA8
    section padded to 16 bytes so that the float4 sections stay aligned; the packed n * 36 layout whenever n is a multiple of 4; the
    Python side (collab.thumbnail_bytes / thumbnail_offsets) says the same."""
    from densemonoslam_amd import capi, collab

    lib = capi.lib
    lib.dms_thumb_block_bytes.restype = ctypes.c_size_t
    lib.dms_thumb_block_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.dms_thumb_block_offsets.restype = None
    lib.dms_thumb_block_offsets.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    for (w, h) in [(640, 480), (320, 240), (1241, 376), (1280, 960), (333, 251)]:
        n = (w // 8) * (h // 8)
        v, nr = ctypes.c_size_t(0), ctypes.c_size_t(0)
        lib.dms_thumb_block_offsets(w, h, ctypes.byref(v), ctypes.byref(nr))
        total = lib.dms_thumb_block_bytes(w, h)
        assert v.value % 16 == 0 and n * 4 <= v.value < n * 4 + 16 and nr.value == v.value + n * 16 and total == nr.value + n * 16
        assert total == collab.thumbnail_bytes(w, h) and (v.value, nr.value) == collab.thumbnail_offsets(w, h)
        if n % 4 == 0:
            assert total == n * 36
    assert lib.dms_thumb_block_bytes(1241, 376) == 29152 + 155 * 47 * 32
