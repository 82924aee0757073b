"""MI355X-native dense tracking + surfel fusion back end (hot path of DenseMonoSLAM).

The product is libdmslam_hip.so (hand-written HIP for gfx950 behind a C ABI, include/*.h);
this package is the Python-side binding used by the tests and the benchmark.
"""
from . import capi  # noqa: F401  (raises ImportError when the HIP library has not been built)
