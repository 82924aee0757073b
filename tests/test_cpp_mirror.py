"""The C++ mirror of the reference classes (densemonoslam_amd/cpp/dmslam.hpp) must compile with a
plain host compiler and link against the C ABI only (no HIP / torch headers)."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstring>
#include "densemonoslam_amd/cpp/dmslam.hpp"
int main() {
  std::printf("%s\n", dms_version());
  // argument validation works without a GPU and reports through the error string
  int rc = dms_createNMap(nullptr, nullptr, nullptr);
  if (rc != DMS_ERR_INVALID_ARG || !std::strstr(dms_last_error(), "null")) return 2;
  try { dms::check(rc, "createNMap"); return 3; } catch (const std::runtime_error&) {}
  dms_fusion_params p;
  dms_fusion_default_params(&p, 640, 480, 528.f, 528.f, 320.f, 240.f);
  return (p.timeDelta == 200 && p.confidence == 10.f && p.maxDepthProcessed == 25.f) ? 0 : 4;
}
"""


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_cpp_mirror_compiles_links_and_runs():
    lib_dir = os.path.join(ROOT, "densemonoslam_amd")
    assert os.path.exists(os.path.join(lib_dir, "libdmslam_hip.so"))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "mirror.cpp")
        exe = os.path.join(td, "mirror")
        with open(src, "w") as f:
            f.write(SRC)
        subprocess.check_call(["g++", "-std=c++14", "-I" + ROOT, src, "-o", exe, "-L" + lib_dir, "-ldmslam_hip",
                               "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert "gfx950" in out.stdout
