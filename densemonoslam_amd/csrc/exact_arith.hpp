// Correctly rounded fp32 quotients, reciprocals and square roots in fewer instructions than the compiler's general sequences
// (v_div_scale x 2 + v_rcp + 5 fma + v_div_fmas + v_div_fixup = 11 for a quotient, 17 for a square root: they carry the scaling for
// operands near the ends of the exponent range and the fix-ups for zero / infinite operands).  The map kernels' fragment stages
// are bound by vector-instruction issue and 4 quotients + 1 square root are half of a fragment's instructions; most of their operands
// sit in ranges where none of that scaffolding can act.  Every function here returns the SAME bits as the IEEE operation on its stated
// domain - by the correction-step theorem (below) for quotients, and by exhaustive comparison on the device for the one-argument
// functions (scripts/micro/exact_arith_check.hip; tests/test_exact_arith_gpu.py runs it).
//
// Correction step (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, 2nd ed., sec. 4.7): with y = RN(1 / b) and q a
// faithful rounding of a / b, r = RN(a - b q) is exact and RN(q + r y) = RN(a / b), absent over / underflow.  q0 = RN(a y) is within
// 1.5 ulp of a / b, so one step gives a / b + (a / b - q0)(1 - b y), i.e. within 2^-24 ulp of a / b before its rounding - faithful -
// and the second step is the theorem's.
#pragma once
#include <hip/hip_runtime.h>

namespace dms {
namespace exact {

// a divisor that many numerators meet (a camera constant; a ray's length shared by its components): d and y = RN(1 / d)
struct Divisor {
  float d, y;
};

// y by the general division: for divisors formed once per launch / row
__device__ __forceinline__ Divisor divisor(float d) { return Divisor{d, 1.0f / d}; }

// RN(a / c.d) for a == 0 (the result is +0 where the division gives -0 for a == -0) or 2^-96 <= |a| finite with 2^-96 <= |a / c.d|
// finite: the remainders are exact there (below, a - d q falls among the subnormals: measured, 10^4 - 10^7 numerators per divisor
// differ).  A NaN stays a NaN, an infinite a gives NaN, a smaller a a quotient of its own size - callers keep those away or add
// something next that swallows them
__device__ __forceinline__ float div(float a, const Divisor& c) {
  float q = a * c.y;
  float r = __builtin_fmaf(-c.d, q, a);
  q = __builtin_fmaf(r, c.y, q);
  r = __builtin_fmaf(-c.d, q, a);
  return __builtin_fmaf(r, c.y, q);
}

// RN(sqrt(a)) for finite a >= 2^-96 (the compiler's sequence minus the scaling of smaller operands and the pass-through of 0 / inf):
// the hardware root is within one unit of the last place; the neighbour whose residual changes sign is taken
__device__ __forceinline__ float sqrt_normal(float a) {
  const float s = __builtin_amdgcn_sqrtf(a);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float ed = __builtin_fmaf(-sd, s, a), eu = __builtin_fmaf(-su, s, a);
  float o = ed <= 0.f ? sd : s;
  o = eu > 0.f ? su : o;
  return o;
}

// RN(1 / s) for s in [1, 4): hardware reciprocal (1 ulp) and one Newton step; exhaustively equal to 1.0f / s over both binades
__device__ __forceinline__ float rcp_1_4(float s) {
  const float y0 = __builtin_amdgcn_rcpf(s);
  const float e = __builtin_fmaf(-s, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}

// the divisor of a vector's length s in [1, 4): {s, RN(1 / s)}
__device__ __forceinline__ Divisor divisor_1_4(float s) { return Divisor{s, rcp_1_4(s)}; }

}  // namespace exact
}  // namespace dms
