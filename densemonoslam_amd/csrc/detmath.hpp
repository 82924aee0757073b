// Deterministic transcendental functions for the fusion kernels.
//
// GLSL leaves exp(), acos(), inversesqrt() precision to the implementation, so the reference
// itself is not bit-defined here.  This build fixes one rule — the functions below, made of
// IEEE add / multiply / divide / sqrt only (no FMA contraction, no libm) — so that the HIP
// kernels and the CPU oracle produce identical bits and surfel association can be required to
// be integer-exact.  Relative accuracy ≈ 2e-7, well inside GLSL's own tolerance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dms {

__host__ __device__ __forceinline__ float det_pow2i(int k) {  // 2^k for k in [-126, 127]
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)(k + 127) << 23;
  return c.f;
}

__host__ __device__ __forceinline__ float det_expf(float x) {
  if (x != x) return x;
  if (x > 88.0f) x = 88.0f;      // callers never exceed 0; saturate instead of inf
  if (x < -87.0f) return 0.0f;   // below 2^-125: flushed to zero by rule
  const float k = rintf(x * 1.44269504088896341f);
  float r = x - k * 0.693145751953125f;
  r = r - k * 1.42860682030941723212e-6f;
  float p = 1.0f / 5040.0f;
  p = p * r + 1.0f / 720.0f;
  p = p * r + 1.0f / 120.0f;
  p = p * r + 1.0f / 24.0f;
  p = p * r + 1.0f / 6.0f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  return p * det_pow2i((int)k);
}

// asin on |z| <= 0.5 (rational approximation of fdlibm's e_asinf form)
__host__ __device__ __forceinline__ float det_asin_core(float z) {
  const float z2 = z * z;
  const float p = z2 * (1.6666586697e-01f + z2 * (-4.2743422091e-02f + z2 * -8.6563630030e-03f));
  const float q = 1.0f + z2 * -7.0662963390e-01f;
  return z + z * (p / q);
}

__host__ __device__ __forceinline__ float det_acosf(float x) {
  if (x != x || x > 1.0f || x < -1.0f) {
    union { uint32_t u; float f; } c;
    c.u = 0x7fc00000u;
    return c.f;
  }
  const float pio2 = 1.57079632679489661923f;
  if (x > -0.5f && x < 0.5f) return pio2 - det_asin_core(x);
  if (x >= 0.5f) {
    const float s = sqrtf((1.0f - x) * 0.5f);
    return 2.0f * det_asin_core(s);
  }
  const float s = sqrtf((1.0f + x) * 0.5f);
  return 2.0f * pio2 - 2.0f * det_asin_core(s);
}

}  // namespace dms
