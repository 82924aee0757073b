// Device self-tests behind the C ABI (include/dmslam.h): the lean correctly-rounded arithmetic of exact_arith.hpp against the compiler's
// IEEE sequences, over whole operand ranges - the evidence that lets the map kernels use it (tests/test_exact_arith_gpu.py).
#include "exact_arith.hpp"
#include "internal.hpp"

namespace dms {

// one thread per 256 consecutive bit patterns starting at `first`; `n` patterns in all
template <int WHAT>
__global__ __launch_bounds__(256) void k_exact_check(unsigned first, unsigned long long n, float d, unsigned long long* __restrict__ bad,
                                                     unsigned* __restrict__ first_bad) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const exact::Divisor c = exact::divisor(d);
  unsigned long long mine = 0;
  unsigned which = 0;
  for (unsigned long long i = t * 256ull; i < t * 256ull + 256ull && i < n; ++i) {
    const unsigned bits = first + (unsigned)i;
    const float a = __uint_as_float(bits);
    float want, got;
    if (WHAT == 0) {
      want = sqrtf(a);
      got = exact::sqrt_normal(a);
    } else if (WHAT == 1) {
      want = 1.0f / a;
      got = exact::rcp_1_4(a);
    } else {
      // the stated domain: a == 0, or 2^-96 <= |a| finite with 2^-96 <= |a / d| finite (the remainder a - d q is exact there)
      if (!(fabsf(a) <= 3.4028234e38f)) continue;
      if (a != 0.f && fabsf(a) < 1.2621774e-29f) continue;
      want = a / d;
      if (a != 0.f && !(fabsf(want) >= 1.2621774e-29f && fabsf(want) <= 3.4028234e38f)) continue;
      got = exact::div(a, c);
      if (want == 0.f && got == 0.f) continue;  // (-0 becomes +0: documented)
    }
    if (__float_as_uint(want) != __float_as_uint(got)) {
      mine += 1;
      which = bits;
    }
  }
  if (mine) {
    atomicAdd(bad, mine);
    atomicMax(first_bad, which);
  }
}

}  // namespace dms

using namespace dms;

extern "C" int dms_exact_arith_selftest(int what, float d, unsigned long long* mismatches, unsigned* first_bad) {
  DMS_REQUIRE(mismatches && what >= 0 && what <= 2, "bad argument");
  DMS_REQUIRE(what != 2 || (d == d && fabsf(d) >= 1.17549435e-38f && fabsf(d) <= 3.4028234e38f), "the divisor must be a normal number");
  unsigned first = 0;
  unsigned long long n = 0;
  if (what == 0) {  // every finite float from 2^-96 up
    first = 0x0F800000u;
    n = 0x7F800000ull - first;
  } else if (what == 1) {  // [1, 4)
    first = 0x3F800000u;
    n = 0x40800000ull - first;
  } else {  // every bit pattern
    first = 0u;
    n = 1ull << 32;
  }
  unsigned long long* d_bad = nullptr;
  DMS_HIP(hipMalloc((void**)&d_bad, 16));
  DMS_HIP(hipMemset(d_bad, 0, 16));
  unsigned* d_first = (unsigned*)(d_bad + 1);
  const unsigned long long threads = (n + 255ull) / 256ull;
  const unsigned blocks = (unsigned)((threads + 255ull) / 256ull);
  if (what == 0)
    hipLaunchKernelGGL(k_exact_check<0>, dim3(blocks), dim3(256), 0, 0, first, n, d, d_bad, d_first);
  else if (what == 1)
    hipLaunchKernelGGL(k_exact_check<1>, dim3(blocks), dim3(256), 0, 0, first, n, d, d_bad, d_first);
  else
    hipLaunchKernelGGL(k_exact_check<2>, dim3(blocks), dim3(256), 0, 0, first, n, d, d_bad, d_first);
  hipError_t e = hipGetLastError();
  unsigned long long host[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(host, d_bad, 16, hipMemcpyDeviceToHost);
  (void)hipFree(d_bad);
  if (e != hipSuccess) return hip_fail(e, "exact-arithmetic self-test", __FILE__, __LINE__);
  *mismatches = host[0];
  if (first_bad) *first_bad = (unsigned)host[1];
  return DMS_OK;
}
