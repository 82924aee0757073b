import os
import sys

import numpy as np
import pytest

# the oracle's OpenMP loops are short: on a many-core host the default (one thread per core) costs more
# in fork/join than it saves
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
# torch brings its own copy of the HIP runtime: whichever of torch / libdmslam_hip.so is loaded first decides which copy the
# process uses, and a process that loads the library first and torch later ends up with two runtimes ("No HIP GPUs are
# available" from the second).  The tests that need both (collab, bench rehearsals) must not depend on the collection order.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gputest_pair():
    """The reference's GPUTest RGB-D pair (elasticfusion/GPUTest/*.png), depth already /5 -> mm
    as the harness does (GPUTest/src/GPUTest.cpp:51-57)."""
    z = np.load(os.path.join(GOLDEN, "gputest_pair.npz"))
    return {
        "rgb1": z["rgb1"],
        "rgb2": z["rgb2"],
        "depth1_raw": z["depth1"],
        "depth2_raw": z["depth2"],
        "depth1": (z["depth1"] // 5).astype(np.uint16),
        "depth2": (z["depth2"] // 5).astype(np.uint16),
        "K": (528.0, 528.0, 320.0, 240.0),  # GPUTest.cpp:150-152
    }


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc

    _orc.set_threads(min(16, os.cpu_count() or 1))
    return _orc
