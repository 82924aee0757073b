/*
 * orc_detmath.h — CPU ORACLE (test infrastructure): the build's fixed rule for the GLSL
 * transcendental functions whose precision OpenGL leaves to the implementation
 * (exp in depth_bilateral.frag:67 and surfels.glsl:45, acos in data.vert:68).  Only IEEE
 * add/multiply/divide/sqrt, evaluated in the order written (compile with -ffp-contract=off),
 * so the result is a pure function of the input bits on any IEEE machine.
 */
#ifndef ORC_DETMATH_H_
#define ORC_DETMATH_H_
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float orc_pow2i(int k) {
  uint32_t u = (uint32_t)(k + 127) << 23;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline float orc_expf(float x) {
  if (x != x) return x;
  if (x > 88.0f) x = 88.0f;
  if (x < -87.0f) return 0.0f;
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
  return p * orc_pow2i((int)k);
}

static inline float orc_asin_core(float z) {
  const float z2 = z * z;
  const float p = z2 * (1.6666586697e-01f + z2 * (-4.2743422091e-02f + z2 * -8.6563630030e-03f));
  const float q = 1.0f + z2 * -7.0662963390e-01f;
  return z + z * (p / q);
}

static inline float orc_acosf(float x) {
  if (x != x || x > 1.0f || x < -1.0f) {
    uint32_t u = 0x7fc00000u;
    float f;
    memcpy(&f, &u, 4);
    return f;
  }
  const float pio2 = 1.57079632679489661923f;
  if (x > -0.5f && x < 0.5f) return pio2 - orc_asin_core(x);
  if (x >= 0.5f) {
    const float s = sqrtf((1.0f - x) * 0.5f);
    return 2.0f * orc_asin_core(s);
  }
  const float s = sqrtf((1.0f + x) * 0.5f);
  return 2.0f * pio2 - 2.0f * orc_asin_core(s);
}
#endif
