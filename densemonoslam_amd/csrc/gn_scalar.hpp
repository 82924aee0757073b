// The scalar section of the tracker: everything the reference does on the host with Eigen between two kernel launches
// (RGBDOdometry.cpp:295-385 for the SO3 pre-alignment, :425-586 for a Gauss-Newton iteration, OdometryProvider.h:35-93),
// here on one GPU lane per block of the resident kernels (or one lane of the solve kernels in `launches` mode).
//
// CANONICAL ARITHMETIC.  Since the cross-pixel sums are order-free (canon.hpp), this section is the only arithmetic
// between two float poses that could still differ between the product and its checker, and a difference of one ulp in
// the float pose is amplified by the next iteration's correspondence search.  It is therefore written as
// ONE fixed sequence of IEEE operations: every multiply-add that is fused is an explicit fma() (the translation unit is
// built with -ffp-contract=off), divisions and square roots are the correctly rounded ones, no hardware estimates, no
// library transcendentals on the paths a tracker update takes.  The same sequence compiled for the host (these are
// __host__ __device__ functions; dms_debug_* entry points in track.hip) and restated in C by the oracle
// (oracle/orc_scalar.c) gives the same bits, which the CPU test suite checks on random systems.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/dmslam.h"

namespace dms {

struct TrackState {
  // prior / current pose (float, as the reference's Eigen float types)
  float Rprev[9], tprev[3], Rprev_inv[9];
  float Rcurr[9], tcurr[3];
  // accumulated incremental transform (RGBDOdometry.cpp:395) and SO3 rotation (:295,301,314)
  double resultRt[16];
  double resultR[9], lastResultR[9];
  float R_lr[9];
  float so3_lastError, so3_lastCount;
  int so3_done, so3_iters;
  // per-iteration projection parameters
  float imageBasis[9], kinv[9], krlr[9];  // SO3 (:321-332)
  float krkinv[9], kt[3];                 // GN  (:427-437)
  int level_done[DMS_NUM_PYRS];
  int iters_run[DMS_NUM_PYRS];
  // side outputs
  float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
  double lastA[36], lastb[6];
  int rejected_jump;
  int sync_timeout;  // 1: a persistent kernel gave up waiting at a grid barrier; 2: no fixed-point range found for a sum (result invalid)
  float out_trans[3], out_rot[9];
  // canonical sums (canon.hpp): column exponents of the ICP and the photometric reduction, carried from iteration to
  // iteration and from level to level of one call; have_E = 0 until the call's first Gauss-Newton reduction
  int E_icp[8], E_rgb[8], have_E;
  int canon_retries;  // reductions of this call that were repeated on a coarser grid (diagnostic)
};

namespace sc {

__host__ __device__ __forceinline__ double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
__host__ __device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__host__ __device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }

// CameraModel::operator()(level) divides in float (types.cuh:115-119); the reciprocals are IEEE divisions
struct KPre {
  double fx, fy, cx, cy, ifx, ify;
};
__host__ __device__ __forceinline__ KPre kpre_of(float fx, float fy, float cx, float cy, int level) {
  const float div = (float)(1 << level);
  KPre k;
  k.fx = (double)(fx / div);
  k.fy = (double)(fy / div);
  k.cx = (double)(cx / div);
  k.cy = (double)(cy / div);
  k.ifx = 1.0 / k.fx;
  k.ify = 1.0 / k.fy;
  return k;
}

// ---- pivoted LDL^T (Eigen's `A.ldlt().solve(b)` semantics, RGBDOdometry.cpp:371,554): diagonal pivoting on |A_kk|, zero
// pivots solved as 0.  Every index is a compile-time constant after unrolling (the matrix lives in registers); the dynamic
// pivot position is resolved by a chain of `if (pp == p)`.
template <typename T, int N>
__host__ __device__ __forceinline__ void ldlt_pivoted(const T (&Ain)[N * N], const T (&b)[N], T (&x)[N], T tiny) {
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
      for (int j = 0; j < k; ++j) akk = fmaT(-A[k * N + j], temp[j], akk);
      A[k * N + k] = akk;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        T v = A[i * N + k];
#pragma unroll
        for (int j = 0; j < k; ++j) v = fmaT(-A[i * N + j], temp[j], v);
        A[i * N + k] = v;
      }
      const T aabs = akk < T(0) ? -akk : akk;
      const bool valid = aabs > T(0);
      if (k == 0 && !valid) {
        all_zero = true;
      } else if (valid) {
#pragma unroll
        for (int i = k + 1; i < N; ++i) A[i * N + k] = A[i * N + k] / akk;
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
    for (int j = 0; j < i; ++j) y[i] = fmaT(-A[i * N + j], y[j], y[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const T d = A[i * N + i];
    const T dabs = d < T(0) ? -d : d;
    y[i] = dabs > tiny ? y[i] / d : T(0);
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i)
#pragma unroll
    for (int j = i + 1; j < N; ++j) y[i] = fmaT(-A[j * N + i], y[j], y[i]);
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

// Fast path for the tracker's normal case: unpivoted LDL^T of a symmetric positive definite matrix, one IEEE reciprocal
// per column.  Returns false (x untouched) unless every pivot is positive and not tiny against the largest diagonal
// entry; the caller then takes the pivoted routine, which defines the result on (near-)singular systems.
template <int N>
__host__ __device__ __forceinline__ bool ldlt_spd(const double (&A)[N * N], const double (&b)[N], double (&x)[N]) {
  double L[N * N], d[N], r[N];
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) dmax = A[i * N + i] > dmax ? A[i * N + i] : dmax;
  const double floor_ = dmax * 1e-11;
  bool ok = dmax > 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double t[N];
    double dk = A[k * N + k];
#pragma unroll
    for (int j = 0; j < k; ++j) {
      t[j] = L[k * N + j] * d[j];
      dk = fmad(-L[k * N + j], t[j], dk);
    }
    d[k] = dk;
    ok = ok && (dk > floor_);
    r[k] = 1.0 / dk;
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      double v = A[i * N + k];
#pragma unroll
      for (int j = 0; j < k; ++j) v = fmad(-L[i * N + j], t[j], v);
      L[i * N + k] = v * r[k];
    }
  }
  if (!ok) return false;
  double y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double v = b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) v = fmad(-L[i * N + j], y[j], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = y[i] * r[i];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) v = fmad(-L[j * N + i], y[j], v);
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = y[i];
  return true;
}

// sin and cos of an angle >= 0.77 as one fixed sequence of IEEE operations (no math library: its results differ between
// the device's and the host's): Cody-Waite reduction by pi / 2 in two parts (fdlibm's pio2_1 / pio2_1t, exact products for
// n < 2^20), then the same fdlibm kernels as above on |r| <= pi / 4; accurate to ~1e-16 for the angles a degenerate system
// can produce, deterministic for all.
__host__ __device__ inline void sincos_canon(double x, double* s_out, double* c_out) {
  const double n = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = fmad(-n, 1.57079632673412561417e+00, x);
  r = fmad(-n, 6.07710050650619224932e-11, r);
  const double z = r * r;
  double a = fmad(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  a = fmad(z, a, 2.75573137070700676789e-06);
  a = fmad(z, a, -1.98412698298579493134e-04);
  a = fmad(z, a, 8.33333333332248946124e-03);
  a = fmad(z, a, -1.66666666666666324348e-01);
  a = fmad(z, a, 1.0);
  double q = fmad(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  q = fmad(z, q, -2.75573143513906633035e-07);
  q = fmad(z, q, 2.48015872894767294178e-05);
  q = fmad(z, q, -1.38888888888741095749e-03);
  q = fmad(z, q, 4.16666666666666019037e-02);
  const double b = fmad(-z, q, 0.5);
  const double sr = a * r, cr = fmad(-z, b, 1.0);
  const int quad = (int)((long long)n & 3);
  const double s = (quad & 1) ? cr : sr, c = (quad & 1) ? sr : cr;
  *s_out = (quad & 2) ? -s : s;
  *c_out = (quad == 1 || quad == 2) ? -c : c;
}

// OdometryProvider::rodrigues (OdometryProvider.h:35-71), row-major 3x3:
//   R = cos(t) I + (1 - cos(t)) r^ r^T + sin(t) [r^]x,  r^ = r / t,  t = |r|.
// For t < 0.77 (every tracker update) the coefficients cos t, (1 - cos t) / t^2 and sin t / t are polynomials in z = t^2
// (the minimax kernels of fdlibm's k_sin.c / k_cos.c, < 1 ulp on |t| <= pi / 4) in Horner form with fma: no square
// root, no division, no argument reduction, R = c I + b r r^T + a [r]x — the reference's matrix to ~2e-16 per entry.
// Larger angles (no tracker update gets there) take the reference's form with sincos_canon.
__host__ __device__ inline void rodrigues(const double* src, double* R) {
  double rx = src[0], ry = src[1], rz = src[2];
  const double z = fmad(rz, rz, fmad(ry, ry, rx * rx));
  if (z < 0.6 && z >= 4.9303806576313238e-32) {  // theta in [DBL_EPSILON, 0.77); below, the reference returns the identity
    double a = fmad(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    a = fmad(z, a, 2.75573137070700676789e-06);
    a = fmad(z, a, -1.98412698298579493134e-04);
    a = fmad(z, a, 8.33333333332248946124e-03);
    a = fmad(z, a, -1.66666666666666324348e-01);
    a = fmad(z, a, 1.0);  // sin t / t
    double q = fmad(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    q = fmad(z, q, -2.75573143513906633035e-07);
    q = fmad(z, q, 2.48015872894767294178e-05);
    q = fmad(z, q, -1.38888888888741095749e-03);
    q = fmad(z, q, 4.16666666666666019037e-02);
    const double b = fmad(-z, q, 0.5);  // (1 - cos t) / t^2
    const double c = fmad(-z, b, 1.0);  // cos t
    const double bx = b * rx, by = b * ry, bz = b * rz;
    R[0] = fmad(bx, rx, c);
    R[1] = fmad(bx, ry, -(a * rz));
    R[2] = fmad(bx, rz, a * ry);
    R[3] = fmad(bx, ry, a * rz);
    R[4] = fmad(by, ry, c);
    R[5] = fmad(by, rz, -(a * rx));
    R[6] = fmad(bx, rz, -(a * ry));
    R[7] = fmad(by, rz, a * rx);
    R[8] = fmad(bz, rz, c);
    return;
  }
  for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
  const double theta = sqrt(z);
  if (theta >= 2.2204460492503131e-16) {
    double s, c;
    sincos_canon(theta, &s, &c);
    const double c1 = 1. - c;
    const double itheta = 1. / theta;
    rx *= itheta;
    ry *= itheta;
    rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) R[k] = (c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k]) + s * rx_[k];
  }
}

// ---- SO3 pre-alignment ------------------------------------------------------------------------------------------------
// K R, K R K^-1 and K^-1 for K = [fx 0 cx; 0 fy cy; 0 0 1] in closed form (RGBDOdometry.cpp:321-332)
__host__ __device__ inline void so3_params(TrackState* st, const KPre& k) {
  const double* R = st->resultR;
  double t[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0 * 3 + j] = fmad(k.cx, R[2 * 3 + j], k.fx * R[0 * 3 + j]);
    t[1 * 3 + j] = fmad(k.cy, R[2 * 3 + j], k.fy * R[1 * 3 + j]);
    t[2 * 3 + j] = R[2 * 3 + j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double h0 = t[i * 3 + 0] * k.ifx, h1 = t[i * 3 + 1] * k.ify;
    st->imageBasis[i * 3 + 0] = (float)h0;
    st->imageBasis[i * 3 + 1] = (float)h1;
    st->imageBasis[i * 3 + 2] = (float)fmad(-h1, k.cy, fmad(-h0, k.cx, t[i * 3 + 2]));
  }
  const double kinv[9] = {k.ifx, 0.0, -(k.cx * k.ifx), 0.0, k.ify, -(k.cy * k.ify), 0.0, 0.0, 1.0};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    st->kinv[i] = (float)kinv[i];
    st->krlr[i] = (float)t[i];
  }
}

// projection parameters of the photometric term for the pose in `resultRt` at camera matrix k (RGBDOdometry.cpp:427-437).
// resultRt is a product of rigid transforms: its inverse is taken in the isometry form [R^T | -R^T t].
__host__ __device__ __forceinline__ void gn_params(const double* resultRt, const KPre& k, float* krkinv, float* kt) {
  double Ri[9], ti[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Ri[i * 3 + j] = resultRt[j * 4 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) ti[i] = -fmad(Ri[i * 3 + 2], resultRt[11], fmad(Ri[i * 3 + 1], resultRt[7], Ri[i * 3 + 0] * resultRt[3]));
  double M[9];  // K * Ri
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M[0 * 3 + j] = fmad(k.cx, Ri[2 * 3 + j], k.fx * Ri[0 * 3 + j]);
    M[1 * 3 + j] = fmad(k.cy, Ri[2 * 3 + j], k.fy * Ri[1 * 3 + j]);
    M[2 * 3 + j] = Ri[2 * 3 + j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double h0 = M[i * 3 + 0] * k.ifx, h1 = M[i * 3 + 1] * k.ify;
    krkinv[i * 3 + 0] = (float)h0;
    krkinv[i * 3 + 1] = (float)h1;
    krkinv[i * 3 + 2] = (float)fmad(-h1, k.cy, fmad(-h0, k.cx, M[i * 3 + 2]));
  }
  kt[0] = (float)fmad(k.cx, ti[2], k.fx * ti[0]);
  kt[1] = (float)fmad(k.cy, ti[2], k.fy * ti[1]);
  kt[2] = (float)ti[2];
}

// float 3x3 helpers of the pose update (every operation rounded, left to right)
__host__ __device__ __forceinline__ void mul3f(const float* a, const float* b, float* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
__host__ __device__ __forceinline__ void mul3vf(const float* a, const float* v, float* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = a[i * 3 + 0] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

// one SO3 update from the sums of an iteration (RGBDOdometry.cpp:334-385); `st` may be the state block in HBM or a copy in LDS
__host__ __device__ inline void so3_solve_core(TrackState* st, const float* sums, float fx, float fy, float cx, float cy, int is_last,
                                               int first_gn_level) {
  float jtj[9], jtr[3];
  int shift = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 4; ++j) {
      const float v = sums[shift++];
      if (j == 3)
        jtr[i] = v;
      else
        jtj[j * 3 + i] = jtj[i * 3 + j] = v;
    }
  const float res0 = sums[9], res1 = sums[10];
  st->so3_iters += 1;
  float err = sqrtf(res0) / res1;
  float cnt = res1;
  bool stop = false;
  if (err < st->so3_lastError && st->so3_lastCount == cnt) {
    stop = true;  // converged
  } else if ((double)err > (double)st->so3_lastError + 0.001) {
    err = st->so3_lastError;  // diverging: roll back
    cnt = st->so3_lastCount;
    for (int i = 0; i < 9; ++i) st->resultR[i] = st->lastResultR[i];
    stop = true;
  }
  st->lastSO3Error = err;
  st->lastSO3Count = cnt;
  if (!stop) {
    st->so3_lastError = err;
    st->so3_lastCount = cnt;
    for (int i = 0; i < 9; ++i) st->lastResultR[i] = st->resultR[i];
    float delta[3];
    ldlt_pivoted<float, 3>(jtj, jtr, delta, 1.0f / 3.402823466e+38F);
    const double dd[3] = {(double)delta[0], (double)delta[1], (double)delta[2]};
    double rotUpdate[9];
    rodrigues(dd, rotUpdate);
    float ru[9], nr[9], lr[9];
    for (int i = 0; i < 9; ++i) {
      ru[i] = (float)rotUpdate[i];
      lr[i] = st->R_lr[i];
    }
    mul3f(ru, lr, nr);
    for (int i = 0; i < 9; ++i) {
      st->R_lr[i] = nr[i];
      st->resultR[i] = (double)nr[i];
    }
  }
  if (stop || is_last) {
    st->so3_done = 1;
    // seed resultRt with the rotation (RGBDOdometry.cpp:397-406) and derive the first GN parameters
    for (int x = 0; x < 3; ++x)
      for (int y = 0; y < 3; ++y) st->resultRt[x * 4 + y] = st->resultR[x * 3 + y];
    double Rt[16];
    for (int i = 0; i < 16; ++i) Rt[i] = st->resultRt[i];
    gn_params(Rt, kpre_of(fx, fy, cx, cy, first_gn_level), st->krkinv, st->kt);
  } else {
    so3_params(st, kpre_of(fx, fy, cx, cy, 2));
  }
}

// ---- Gauss-Newton iteration ------------------------------------------------------------------------------------------
struct SolveArgs {
  int icp, rgb, rgbOnly;
  float icpWeight;
  int level, first_iter, next_level, level_below;
  float fx, fy, cx, cy;
};

// State of one Gauss-Newton level that evolves from iteration to iteration.  The launch path keeps it in registers for one
// k_gn_solve; the resident path keeps it in LDS for a whole level.
struct GnLocal {
  double resultRt[16];
  float Rprev[9], tprev[3], Rprev_inv[9];  // constant during the loop
  float Rcurr[9], tcurr[3];
  float krkinv[9], kt[3];
  float lastRGBError, lastRGBCount, lastICPError, lastICPCount;
  int iters_run;
  double lastA[36], lastb[6];
};

// Entry k of the combined system from entry k of the two reductions (RGBDOdometry.cpp:531-552: A = A_rgb + w^2 A_icp,
// b = b_rgb + w b_icp; one fma per entry).  k runs over the 21 + 6 unique entries, rows of (A | b): the b entries are
// k = 6, 12, 17, 21, 24, 26.  The resident kernels evaluate this in the lanes that hold the totals (lane k), every other
// path through gn_step_core below: the same operation either way.
__host__ __device__ __forceinline__ constexpr bool comb_is_b(int k) { return k == 6 || k == 12 || k == 17 || k == 21 || k == 24 || k == 26; }
__host__ __device__ __forceinline__ double comb_entry(int k, bool icp, bool rgb, float icpWeight, float v_icp, float v_rgb) {
  const double w = (double)icpWeight;
  const double ww = w * w;
  const double vi = icp ? (double)v_icp : 0.0, vr = rgb ? (double)v_rgb : 0.0;
  if (icp && rgb) return comb_is_b(k) ? fmad(w, vi, vr) : fmad(ww, vi, vr);
  return icp ? vi : vr;
}

// One Gauss-Newton update (RGBDOdometry.cpp:472-585) from the combined system `comb` (27 entries, comb_entry): LDL^T in fp64,
// se(3) update of resultRt, new float pose, projection parameters for the camera matrix `kpre` (of q.next_level).
// res0 / res1: the ICP residual sum and inlier count (side outputs).
// `side`: store the side outputs lastA / lastb / last*Error / last*Count (only the values of a level's last iteration are read).
__host__ __device__ __forceinline__ void gn_step_combined(GnLocal& L, const double* comb, float res0, float res1, int rgbSize, int sigma,
                                                          const SolveArgs& q, const KPre& kpre, bool side = true) {
  const float residual[2] = {q.icp ? res0 : 0.f, q.icp ? res1 : 0.f};
  double A[36], b[6], x[6];
  {
    int shift = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 7; ++j) {
        const double v = comb[shift++];
        if (j == 6)
          b[i] = v;
        else
          A[j * 6 + i] = A[i * 6 + j] = v;
      }
  }
  if (!ldlt_spd<6>(A, b, x)) ldlt_pivoted<double, 6>(A, b, x, 1.0 / 1.7976931348623157e308);

  // OdometryProvider::computeUpdateSE3 (OdometryProvider.h:73-93): resultRt = [exp(x)] * resultRt, both with last row (0 0 0 1)
  const double rvec[3] = {x[3], x[4], x[5]};
  double R[9];
  rodrigues(rvec, R);
  double nr[16];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double v = fmad(R[i * 3 + 2], L.resultRt[2 * 4 + j], fmad(R[i * 3 + 1], L.resultRt[1 * 4 + j], R[i * 3 + 0] * L.resultRt[0 * 4 + j]));
      if (j == 3) v += x[i];
      nr[i * 4 + j] = v;
    }
  }
  nr[12] = 0.0;
  nr[13] = 0.0;
  nr[14] = 0.0;
  nr[15] = 1.0;

  // rgbOdom = float(resultRt); currentT = [Rprev|tprev] * rgbOdom^-1 with the isometry inverse (:573-585), in float
  float Ro[9], to[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Ro[i * 3 + j] = (float)nr[i * 4 + j];
    to[i] = (float)nr[i * 4 + 3];
  }
  float RoT[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) RoT[i * 3 + j] = Ro[j * 3 + i];
  float ti[3];
  mul3vf(RoT, to, ti);
  ti[0] = -ti[0];
  ti[1] = -ti[1];
  ti[2] = -ti[2];
  float Rprev[9], Rc[9], tc[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rprev[i] = L.Rprev[i];
  mul3f(Rprev, RoT, Rc);
  mul3vf(Rprev, ti, tc);

  L.iters_run += 1;
  if (side) {
    L.lastRGBError = (float)(sqrt((double)sigma) / (double)rgbSize);
    L.lastRGBCount = (float)rgbSize;
    L.lastICPError = sqrtf(residual[0]) / residual[1];
    L.lastICPCount = residual[1];
#pragma unroll
    for (int i = 0; i < 36; ++i) L.lastA[i] = A[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) L.lastb[i] = b[i];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) L.resultRt[i] = nr[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) L.Rcurr[i] = Rc[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) L.tcurr[i] = tc[i] + L.tprev[i];
  gn_params(nr, kpre, L.krkinv, L.kt);
}

// The same from the sums of the two reductions (the launch-per-phase path, the host build of the CPU tests)
__host__ __device__ __forceinline__ void gn_step_core(GnLocal& L, const float* s_icp, const float* s_rgb, int rgbSize, int sigma,
                                                      const SolveArgs& q, const KPre& kpre, bool side = true) {
  double comb[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) comb[k] = comb_entry(k, q.icp != 0, q.rgb != 0, q.icpWeight, q.icp ? s_icp[k] : 0.f, q.rgb ? s_rgb[k] : 0.f);
  gn_step_combined(L, comb, q.icp ? s_icp[27] : 0.f, q.icp ? s_icp[28] : 0.f, rgbSize, sigma, q, kpre, side);
}

// sigma as the reference computes it (RGBDOdometry.cpp:464, precedence quirk kept)
__host__ __device__ __forceinline__ float sigma_val(int sigma, int rgbSize) {
  const float q = (float)sigma / (float)rgbSize;
  const int arg = (q == 0.f) ? 1 : rgbSize;
  return (float)sqrt((double)arg);
}
__host__ __device__ __forceinline__ bool rgbonly_break(int sigma, int rgbSize, float lastRGBError) {
  return sqrt((double)sigma) / (double)rgbSize > (double)lastRGBError;
}

}  // namespace sc
}  // namespace dms
