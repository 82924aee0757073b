// Small dense linear algebra used by the on-device Gauss-Newton solve: the pieces the
// reference does on the host with Eigen (RGBDOdometry.cpp:295-585, OdometryProvider.h:35-93),
// restated for a single GPU lane in fp64 (fp32 where the reference uses float).
#pragma once
#include <hip/hip_runtime.h>

namespace dms {
namespace sm {

// 3×3 inverse through cofactors and one reciprocal of the determinant (row-major).
template <typename T>
__host__ __device__ inline void inv3(const T* m, T* o) {
  const T c00 = m[4] * m[8] - m[5] * m[7];
  const T c01 = m[5] * m[6] - m[3] * m[8];
  const T c02 = m[3] * m[7] - m[4] * m[6];
  const T det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const T id = T(1) / det;
  o[0] = c00 * id;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

template <typename T>
__host__ __device__ inline void mul3(const T* a, const T* b, T* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

template <typename T>
__host__ __device__ inline void mul3v(const T* a, const T* v, T* o) {
  for (int i = 0; i < 3; ++i) o[i] = a[i * 3 + 0] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

__host__ __device__ inline void mul4(const double* a, const double* b, double* o) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = a[i * 4 + 0] * b[0 * 4 + j];
      s += a[i * 4 + 1] * b[1 * 4 + j];
      s += a[i * 4 + 2] * b[2 * 4 + j];
      s += a[i * 4 + 3] * b[3 * 4 + j];
      o[i * 4 + j] = s;
    }
}

// general 4×4 inverse by cofactor expansion (row-major)
template <typename T>
__host__ __device__ inline void inv4t(const T* m, T* o) {
  T inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const T det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const T id = T(1) / det;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = inv[i] * id;
}
// c = a * b for row-major 4 x 4 float matrices, every element ((a0 b0 + a1 b1) + a2 b2) + a3 b3 with every operation rounded
// (the order of oracle/orc_ferns.py _mul44 and of the map-merge kernels; the library is built with -ffp-contract=off)
inline void mul44_host(const float* a, const float* b, float* o) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = a[i * 4 + 0] * b[0 * 4 + j];
      s = s + a[i * 4 + 1] * b[1 * 4 + j];
      s = s + a[i * 4 + 2] * b[2 * 4 + j];
      s = s + a[i * 4 + 3] * b[3 * 4 + j];
      o[i * 4 + j] = s;
    }
}
__host__ __device__ inline void inv4(const double* m, double* o) { inv4t<double>(m, o); }

// Pivoted LDLT solve of a symmetric N×N system (diagonal pivoting on |A_kk|, zero pivots
// solved as 0): the algorithm behind `A.ldlt().solve(b)` at RGBDOdometry.cpp:371,554.
// `A` (N*N), `temp`/`y` (N each) and `perm` (N) are caller-provided scratch: on the GPU they are
// LDS arrays, because the pivoting indexes them dynamically and private arrays with dynamic
// indices live in scratch memory (hundreds of cycles per access on the single solving lane).
template <typename T, int N>
__host__ __device__ inline void ldlt_solve_ws(const T* Ain, const T* b, T* x, T tiny, T* A, T* temp, T* y, int* perm) {
  for (int i = 0; i < N * N; ++i) A[i] = Ain[i];
  bool all_zero = false;
  for (int k = 0; k < N; ++k) {
    int p = k;
    T best = A[k * N + k] < T(0) ? -A[k * N + k] : A[k * N + k];
    for (int i = k + 1; i < N; ++i) {
      const T v = A[i * N + i] < T(0) ? -A[i * N + i] : A[i * N + i];
      if (v > best) {
        best = v;
        p = i;
      }
    }
    perm[k] = p;
    if (p != k) {
      for (int j = 0; j < N; ++j) {
        const T t = A[k * N + j];
        A[k * N + j] = A[p * N + j];
        A[p * N + j] = t;
      }
      for (int i = 0; i < N; ++i) {
        const T t = A[i * N + k];
        A[i * N + k] = A[i * N + p];
        A[i * N + p] = t;
      }
    }
    // lower-triangular update: column k below the diagonal
    for (int j = 0; j < k; ++j) temp[j] = A[j * N + j] * A[k * N + j];
    T akk = A[k * N + k];
    for (int j = 0; j < k; ++j) akk -= A[k * N + j] * temp[j];
    A[k * N + k] = akk;
    for (int i = k + 1; i < N; ++i) {
      T v = A[i * N + k];
      for (int j = 0; j < k; ++j) v -= A[i * N + j] * temp[j];
      A[i * N + k] = v;
    }
    const T aabs = akk < T(0) ? -akk : akk;
    const bool valid = aabs > T(0);
    if (k == 0 && !valid) {
      for (int j = 0; j < N; ++j) perm[j] = j;
      all_zero = true;
      break;
    }
    if (valid)
      for (int i = k + 1; i < N; ++i) A[i * N + k] /= akk;
  }
  for (int i = 0; i < N; ++i) y[i] = b[i];
  if (all_zero) {
    for (int i = 0; i < N; ++i) x[i] = T(0);
    return;
  }
  for (int k = 0; k < N; ++k)
    if (perm[k] != k) {
      const T t = y[k];
      y[k] = y[perm[k]];
      y[perm[k]] = t;
    }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * N + j] * y[j];
  for (int i = 0; i < N; ++i) {
    const T d = A[i * N + i];
    const T dabs = d < T(0) ? -d : d;
    y[i] = dabs > tiny ? y[i] / d : T(0);
  }
  for (int i = N - 1; i >= 0; --i)
    for (int j = i + 1; j < N; ++j) y[i] -= A[j * N + i] * y[j];
  for (int k = N - 1; k >= 0; --k)
    if (perm[k] != k) {
      const T t = y[k];
      y[k] = y[perm[k]];
      y[perm[k]] = t;
    }
  for (int i = 0; i < N; ++i) x[i] = y[i];
}

// Same algorithm with every index a compile-time constant, so the matrix lives in registers on
// the solving lane (no scratch, no LDS round trips).  The dynamic pivot position is resolved by a
// chain of `if (p == pivot)` over fully unrolled loops; `swap_rc<K>` exchanges rows and columns K
// and P of the full symmetric storage exactly as the loop version does.
template <typename T, int N>
__host__ __device__ __forceinline__ void ldlt_solve_reg(const T (&Ain)[N * N], const T (&b)[N], T (&x)[N], T tiny) {
#pragma clang fp contract(fast)  // scalar section of the tracker: fused multiply-adds (see gn_step_core)
  T A[N * N];
#pragma unroll
  for (int i = 0; i < N * N; ++i) A[i] = Ain[i];
  int perm[N];
  bool all_zero = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (!all_zero) {
      int p = k;
      T best = A[k * N + k] < T(0) ? -A[k * N + k] : A[k * N + k];
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const T v = A[i * N + i] < T(0) ? -A[i * N + i] : A[i * N + i];
        if (v > best) {
          best = v;
          p = i;
        }
      }
      perm[k] = p;
#pragma unroll
      for (int pp = k + 1; pp < N; ++pp) {
        if (pp == p) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const T t = A[k * N + j];
            A[k * N + j] = A[pp * N + j];
            A[pp * N + j] = t;
          }
#pragma unroll
          for (int i = 0; i < N; ++i) {
            const T t = A[i * N + k];
            A[i * N + k] = A[i * N + pp];
            A[i * N + pp] = t;
          }
        }
      }
      T temp[N];
#pragma unroll
      for (int j = 0; j < k; ++j) temp[j] = A[j * N + j] * A[k * N + j];
      T akk = A[k * N + k];
#pragma unroll
      for (int j = 0; j < k; ++j) akk -= A[k * N + j] * temp[j];
      A[k * N + k] = akk;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        T v = A[i * N + k];
#pragma unroll
        for (int j = 0; j < k; ++j) v -= A[i * N + j] * temp[j];
        A[i * N + k] = v;
      }
      const T aabs = akk < T(0) ? -akk : akk;
      const bool valid = aabs > T(0);
      if (k == 0 && !valid) {
        all_zero = true;
      } else if (valid) {
#pragma unroll
        for (int i = k + 1; i < N; ++i) A[i * N + k] /= akk;
      }
    }
  }
  if (all_zero) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = T(0);
    return;
  }
  T y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = b[i];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int pp = k + 1; pp < N; ++pp)
      if (perm[k] == pp) {
        const T t = y[k];
        y[k] = y[pp];
        y[pp] = t;
      }
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) y[i] -= A[i * N + j] * y[j];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const T d = A[i * N + i];
    const T dabs = d < T(0) ? -d : d;
    y[i] = dabs > tiny ? y[i] / d : T(0);
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i)
#pragma unroll
    for (int j = i + 1; j < N; ++j) y[i] -= A[j * N + i] * y[j];
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
#pragma unroll
    for (int pp = k + 1; pp < N; ++pp)
      if (perm[k] == pp) {
        const T t = y[k];
        y[k] = y[pp];
        y[pp] = t;
      }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = y[i];
}

// 1 / d for the unpivoted solve: the hardware estimate and two Newton steps (the first four operations of the
// IEEE division sequence, without its scaling and final correction): within ~1e-16 of the quotient for normal d
__host__ __device__ __forceinline__ double rcp_newton(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(e, r, r);
#else
  return 1.0 / d;
#endif
}
__host__ __device__ __forceinline__ float rcp_newton(float d) { return 1.0f / d; }

// Fast path for the tracker's normal case: an unpivoted LDL^T of a symmetric positive definite
// matrix, all in registers, one reciprocal (estimate + two Newton steps) per column.  Returns false (x untouched) unless every
// pivot is positive and not tiny against the largest diagonal entry, i.e. unless the matrix is
// safely positive definite; the caller then falls back to the pivoted routine above, which is what
// defines the result on (near-)singular systems.  On accepted systems the two solutions agree to
// cond(A) * 1e-16.
template <typename T, int N>
__host__ __device__ __forceinline__ bool ldlt_solve_spd(const T (&A)[N * N], const T (&b)[N], T (&x)[N]) {
#pragma clang fp contract(fast)  // scalar section of the tracker: fused multiply-adds (see gn_step_core)
  T L[N * N], d[N], r[N];
  T dmax = T(0);
#pragma unroll
  for (int i = 0; i < N; ++i) dmax = A[i * N + i] > dmax ? A[i * N + i] : dmax;
  const T floor_ = dmax * T(1e-11);
  bool ok = dmax > T(0);
#pragma unroll
  for (int k = 0; k < N; ++k) {
    T t[N];
    T dk = A[k * N + k];
#pragma unroll
    for (int j = 0; j < k; ++j) {
      t[j] = L[k * N + j] * d[j];
      dk -= L[k * N + j] * t[j];
    }
    d[k] = dk;
    ok = ok && (dk > floor_);
    r[k] = rcp_newton(dk);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      T v = A[i * N + k];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= L[i * N + j] * t[j];
      L[i * N + k] = v * r[k];
    }
  }
  if (!ok) return false;
  T y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    T v = b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) v -= L[i * N + j] * y[j];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] *= r[i];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    T v = y[i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) v -= L[j * N + i] * y[j];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = y[i];
  return true;
}

template <typename T, int N>
__host__ __device__ inline void ldlt_solve(const T* Ain, const T* b, T* x, T tiny) {
  T A[N * N], temp[N], y[N];
  int perm[N];
  ldlt_solve_ws<T, N>(Ain, b, x, tiny, A, temp, y, perm);
}

// reference OdometryProvider::rodrigues (OdometryProvider.h:35-71), row-major 3×3:
//   R = cos(t) I + (1 - cos(t)) r^ r^T + sin(t) [r^]x,  r^ = r / t,  t = |r|
// For t < 0.78 (every Gauss-Newton update) the three coefficients cos(t), (1 - cos(t)) / t^2 and sin(t) / t are taken as
// polynomials in z = t^2 (the minimax kernels of fdlibm's k_sin.c / k_cos.c, < 1 ulp on |t| <= pi/4), which needs no
// square root, no division and no argument reduction: R = c I + b r r^T + a [r]x, the same matrix to ~2e-16 per entry
// before the pose is rounded to float.  Larger angles take the reference's form.
__host__ __device__ inline void rodrigues(const double* src, double* R) {
#pragma clang fp contract(fast)  // scalar section of the tracker: fused multiply-adds (see gn_step_core)
  double rx = src[0], ry = src[1], rz = src[2];
  const double z = rx * rx + ry * ry + rz * rz;
  if (z < 0.6 && z >= 4.9303806576313238e-32) {  // theta in [DBL_EPSILON, 0.77): below, the reference returns the identity
    const double a = 1.0 + z * (-1.66666666666666324348e-01 +
                                z * (8.33333333332248946124e-03 +
                                     z * (-1.98412698298579493134e-04 +
                                          z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
    const double q = 4.16666666666666019037e-02 +
                     z * (-1.38888888888741095749e-03 +
                          z * (2.48015872894767294178e-05 +
                               z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double b = 0.5 - z * q;  // (1 - cos t) / t^2
    const double c = 1.0 - z * b;  // cos t
    const double bx = b * rx, by = b * ry, bz = b * rz;
    R[0] = c + bx * rx;
    R[1] = bx * ry - a * rz;
    R[2] = bx * rz + a * ry;
    R[3] = bx * ry + a * rz;
    R[4] = c + by * ry;
    R[5] = by * rz - a * rx;
    R[6] = bx * rz - a * ry;
    R[7] = by * rz + a * rx;
    R[8] = c + bz * rz;
    return;
  }
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) R[k] = I[k];
  const double theta = sqrt(z);
  if (theta >= 2.2204460492503131e-16) {
    double s, c;
    sincos(theta, &s, &c);  // one range reduction for both
    const double c1 = 1. - c;
    const double itheta = theta ? 1. / theta : 0.;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
  }
}

}  // namespace sm
}  // namespace dms
