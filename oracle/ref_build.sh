#!/bin/bash
# Test infrastructure: builds oracle/_ref/libref_reduce.so = the REFERENCE's own tracking kernels
# (elasticfusion/Core/src/Cuda/reduce.cu: icpStep, rgbStep, so3Step, computeRgbResidual, with its
# containers/ and headers), compiled for gfx950 from the sources where they lie under /root/reference.
#
# How: the image's own CUDA->HIP translator (/opt/rocm/bin/hipify-perl, an identifier-renaming script:
# cudaMalloc -> hipMalloc, <cuda_runtime.h> -> <hip/hip_runtime.h>, ...) writes the renamed text into
# oracle/_ref/src/ (git-ignored build output, never committed), and hipcc compiles that together with
# ref_reduce_harness.cpp (ours: extern "C" wrappers that upload host arrays into the reference's
# DeviceArray2D and call the reference's functions with the reference's argument lists).
# No reference source is edited by hand, no header / library / tool is written as a stand-in.
# The recipe's own interventions, all of them:
#   * one header rename hipify-perl's table lacks (vector_functions.h -> hip/hip_vector_types.h, the HIP header that
#     declares make_float3 and friends; its sibling vector_types.h IS in the table and maps to the same file);
#   * -D__CUDACC__ (types.cuh:57 includes Eigen unless the CUDA compiler is compiling: hipcc is that compiler here);
#   * -D__CUDA_ARCH__=350 (reduce.cu:56-84 otherwise defines its own pre-Kepler __shfl_down / __ldg emulation;
#     350 selects the hardware shuffles, as every supported NVIDIA part does);
#   * -Wno-c++11-narrowing (reduce.cu:342,618 initialise a float from a bool in a braced list: nvcc accepts, clang warns-as-error);
#   * -ffp-contract=off: a compiler choice either way (nvcc contracts by default, at places of its choosing); without
#     contraction the per-pixel arithmetic is the source's, operation by operation, which is what the oracle's plain
#     (non-fused) mode restates, so decisions (inliers, correspondences) can be compared exactly.
# Not built: cudafuncs.cu (its imageBGRToIntensity samples a legacy texture reference; gfx950 has no image
# instructions and HIP marks tex2D unavailable there) - unbuildable here, its restatement stays unpinned.
# The wavefront is 64 wide: the reference's block reduction is written against warpSize and is correct for any launch
# whose thread count is a multiple of it (the harness's callers use 128 / 256; the reference's own 160 for so3Step is
# not a multiple of 64 and would drop the last half wave).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${DMS_REFERENCE_ROOT:-/root/reference}/elasticfusion/Core/src/Cuda"
OUT="$HERE/_ref"
if [ ! -f "$REF/reduce.cu" ]; then
  echo "ref_build.sh: $REF/reduce.cu not found (no reference here): nothing built" >&2
  exit 0
fi
mkdir -p "$OUT/src/containers"
for f in reduce.cu cudafuncs.cuh types.cuh convenience.cuh operators.cuh \
         containers/device_array.hpp containers/device_array_impl.hpp containers/device_memory.cpp \
         containers/device_memory.hpp containers/device_memory_impl.hpp containers/kernel_containers.hpp; do
  hipify-perl -quiet-warnings "$REF/$f" > "$OUT/src/$f" 2>/dev/null
done
sed -i 's|<vector_functions.h>|<hip/hip_vector_types.h>|' "$OUT/src/operators.cuh"
FLAGS="-x hip --offload-arch=gfx950 -O2 -fPIC -ffp-contract=off -fno-fast-math -D__CUDACC__ -D__CUDA_ARCH__=350 \
 -Wno-c++11-narrowing -Wno-unused-value -Wno-unused-result -include hip/hip_runtime.h -I$OUT/src"
hipcc $FLAGS -c "$OUT/src/reduce.cu" -o "$OUT/reduce.o"
hipcc $FLAGS -c "$OUT/src/containers/device_memory.cpp" -o "$OUT/device_memory.o"
hipcc $FLAGS -c "$HERE/ref_reduce_harness.cpp" -o "$OUT/harness.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libref_reduce.so" "$OUT/reduce.o" "$OUT/device_memory.o" "$OUT/harness.o"
# second variant: the same sources with the compiler's DEFAULT contraction of device code (what nvcc's default -fmad=true is to
# the reference's own build): multiply-adds fused where the compiler sees them.  The product's resident tracker evaluates its
# rows with every multiply-add chain fused (pixel_ops.hpp madd<true>), restated by the oracle's fused mode; this library shows
# how a compiler's own choice of fusions compares with that (tests/golden/make_ref_reduce_golden.py records it).
FLAGS_FMA="${FLAGS/-ffp-contract=off/-ffp-contract=fast}"
hipcc $FLAGS_FMA -c "$OUT/src/reduce.cu" -o "$OUT/reduce_fma.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libref_reduce_fma.so" "$OUT/reduce_fma.o" "$OUT/device_memory.o" "$OUT/harness.o"
rm -rf "$OUT"/*.o "$OUT/src"   # only the library stays: no reference text is left in the tree
echo "built $OUT/libref_reduce.so $OUT/libref_reduce_fma.so"
