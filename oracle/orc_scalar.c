/*
 * orc_scalar.c — CPU ORACLE (test infrastructure; see the header of orc_track.c).
 *
 * The scalar section of the tracker — what the reference does on the host with Eigen between two kernel launches
 * (RGBDOdometry.cpp:295-385, :425-586, OdometryProvider.h:35-93) — restated in the CANONICAL operation order of the
 * product (densemonoslam_amd/csrc/gn_scalar.hpp): one fixed sequence of IEEE operations, every fused multiply-add
 * explicit, so that both sides produce the same bits from the same sums.  Why that is needed: a difference of one ulp in
 * the float pose of an iteration changes a few of the next iteration's 300 000 correspondence decisions, and the
 * Gauss-Newton loop carries such a perturbation to the end of the call (scripts/sum_order_control.py measures it).
 * orc_odometry.c keeps the independent restatement (Eigen-like pivoted LDLT in outer-product form, Rodrigues through the
 * math library, general matrix inverses) as solve mode 0; tests/test_oracle_cpu.py requires the two to agree to ~1e-12.
 *
 * Built with -ffp-contract=off: nothing is fused unless fma() says so.
 */
#include <float.h>
#include <math.h>
#include <string.h>

#include "orc_scalar.h"

static inline double fmad(double a, double b, double c) { return fma(a, b, c); }

orc_kpre orc_kpre_of(float fx, float fy, float cx, float cy, int level) { /* CameraModel::operator()(level), types.cuh:115-119 */
  const float div = (float)(1 << level);
  orc_kpre k;
  k.fx = (double)(fx / div);
  k.fy = (double)(fy / div);
  k.cx = (double)(cx / div);
  k.cy = (double)(cy / div);
  k.ifx = 1.0 / k.fx;
  k.ify = 1.0 / k.fy;
  return k;
}

/* `A.ldlt().solve(b)` (RGBDOdometry.cpp:371,554): diagonal pivoting on |A_kk|, zero pivots solved as 0 */
#define LDLT_PIVOTED(NAME, T, FMA)                                                                      \
  void NAME(int n, const T* Ain, const T* b, T* x, T tiny) {                                            \
    T A[36], temp[6], y[6];                                                                             \
    int perm[6];                                                                                        \
    for (int i = 0; i < n * n; ++i) A[i] = Ain[i];                                                      \
    for (int k = 0; k < n; ++k) {                                                                       \
      int p = k;                                                                                        \
      T best = A[k * n + k] < (T)0 ? -A[k * n + k] : A[k * n + k];                                      \
      for (int i = k + 1; i < n; ++i) {                                                                 \
        const T v = A[i * n + i] < (T)0 ? -A[i * n + i] : A[i * n + i];                                 \
        if (v > best) {                                                                                 \
          best = v;                                                                                     \
          p = i;                                                                                        \
        }                                                                                               \
      }                                                                                                 \
      perm[k] = p;                                                                                      \
      if (p != k) {                                                                                     \
        for (int j = 0; j < n; ++j) {                                                                   \
          const T t = A[k * n + j];                                                                     \
          A[k * n + j] = A[p * n + j];                                                                  \
          A[p * n + j] = t;                                                                             \
        }                                                                                               \
        for (int i = 0; i < n; ++i) {                                                                   \
          const T t = A[i * n + k];                                                                     \
          A[i * n + k] = A[i * n + p];                                                                  \
          A[i * n + p] = t;                                                                             \
        }                                                                                               \
      }                                                                                                 \
      for (int j = 0; j < k; ++j) temp[j] = A[j * n + j] * A[k * n + j];                                \
      T akk = A[k * n + k];                                                                             \
      for (int j = 0; j < k; ++j) akk = FMA(-A[k * n + j], temp[j], akk);                               \
      A[k * n + k] = akk;                                                                               \
      for (int i = k + 1; i < n; ++i) {                                                                 \
        T v = A[i * n + k];                                                                             \
        for (int j = 0; j < k; ++j) v = FMA(-A[i * n + j], temp[j], v);                                 \
        A[i * n + k] = v;                                                                               \
      }                                                                                                 \
      const T aabs = akk < (T)0 ? -akk : akk;                                                           \
      const int valid = aabs > (T)0;                                                                    \
      if (k == 0 && !valid) {                                                                           \
        for (int i = 0; i < n; ++i) x[i] = (T)0;                                                        \
        return;                                                                                         \
      }                                                                                                 \
      if (valid)                                                                                        \
        for (int i = k + 1; i < n; ++i) A[i * n + k] = A[i * n + k] / akk;                              \
    }                                                                                                   \
    for (int i = 0; i < n; ++i) y[i] = b[i];                                                            \
    for (int k = 0; k < n; ++k)                                                                         \
      if (perm[k] != k) {                                                                               \
        const T t = y[k];                                                                               \
        y[k] = y[perm[k]];                                                                              \
        y[perm[k]] = t;                                                                                 \
      }                                                                                                 \
    for (int i = 0; i < n; ++i)                                                                         \
      for (int j = 0; j < i; ++j) y[i] = FMA(-A[i * n + j], y[j], y[i]);                                \
    for (int i = 0; i < n; ++i) {                                                                       \
      const T d = A[i * n + i];                                                                         \
      const T dabs = d < (T)0 ? -d : d;                                                                 \
      y[i] = dabs > tiny ? y[i] / d : (T)0;                                                             \
    }                                                                                                   \
    for (int i = n - 1; i >= 0; --i)                                                                    \
      for (int j = i + 1; j < n; ++j) y[i] = FMA(-A[j * n + i], y[j], y[i]);                            \
    for (int k = n - 1; k >= 0; --k)                                                                    \
      if (perm[k] != k) {                                                                               \
        const T t = y[k];                                                                               \
        y[k] = y[perm[k]];                                                                              \
        y[perm[k]] = t;                                                                                 \
      }                                                                                                 \
    for (int i = 0; i < n; ++i) x[i] = y[i];                                                            \
  }
LDLT_PIVOTED(orc_scalar_ldlt_pivoted_d, double, fma)
LDLT_PIVOTED(orc_scalar_ldlt_pivoted_f, float, fmaf)

/* unpivoted LDL^T of a safely positive definite 6x6 system; returns 0 (x untouched) otherwise */
int orc_scalar_ldlt_spd6(const double* A, const double* b, double* x) {
  enum { N = 6 };
  double L[N * N], d[N], r[N];
  double dmax = 0.0;
  for (int i = 0; i < N; ++i) dmax = A[i * N + i] > dmax ? A[i * N + i] : dmax;
  const double floor_ = dmax * 1e-11;
  int ok = dmax > 0.0;
  for (int k = 0; k < N; ++k) {
    double t[N];
    double dk = A[k * N + k];
    for (int j = 0; j < k; ++j) {
      t[j] = L[k * N + j] * d[j];
      dk = fmad(-L[k * N + j], t[j], dk);
    }
    d[k] = dk;
    ok = ok && (dk > floor_);
    r[k] = 1.0 / dk;
    for (int i = k + 1; i < N; ++i) {
      double v = A[i * N + k];
      for (int j = 0; j < k; ++j) v = fmad(-L[i * N + j], t[j], v);
      L[i * N + k] = v * r[k];
    }
  }
  if (!ok) return 0;
  double y[N];
  for (int i = 0; i < N; ++i) {
    double v = b[i];
    for (int j = 0; j < i; ++j) v = fmad(-L[i * N + j], y[j], v);
    y[i] = v;
  }
  for (int i = 0; i < N; ++i) y[i] = y[i] * r[i];
  for (int i = N - 1; i >= 0; --i) {
    double v = y[i];
    for (int j = i + 1; j < N; ++j) v = fmad(-L[j * N + i], y[j], v);
    y[i] = v;
  }
  for (int i = 0; i < N; ++i) x[i] = y[i];
  return 1;
}

/* sin and cos of an angle >= 0.77 as one fixed sequence of IEEE operations (no math library: its results differ between */
/* the device's and the host's): Cody-Waite reduction by pi / 2 in two parts (fdlibm's pio2_1 / pio2_1t, exact products for */
/* n < 2^20), then the same fdlibm kernels as above on |r| <= pi / 4; accurate to ~1e-16 for the angles a degenerate system */
/* can produce, deterministic for all. */
static void sincos_canon(double x, double* s_out, double* c_out) {
  const double n = rint(x * 6.36619772367581382433e-01);
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

/* OdometryProvider::rodrigues (OdometryProvider.h:35-71) with polynomial coefficients for |r| < 0.77 (fdlibm k_sin / k_cos) */
void orc_scalar_rodrigues(const double* src, double* R) {
  double rx = src[0], ry = src[1], rz = src[2];
  const double z = fmad(rz, rz, fmad(ry, ry, rx * rx));
  if (z < 0.6 && z >= 4.9303806576313238e-32) {
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
    const double c = fmad(-z, b, 1.0);
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

/* K R, K R K^-1 and K^-1 in closed form (RGBDOdometry.cpp:321-332) */
void orc_scalar_so3_params(const double* R, const orc_kpre* k, float* imageBasis, float* kinv, float* krlr) {
  double t[9];
  for (int j = 0; j < 3; ++j) {
    t[0 * 3 + j] = fmad(k->cx, R[2 * 3 + j], k->fx * R[0 * 3 + j]);
    t[1 * 3 + j] = fmad(k->cy, R[2 * 3 + j], k->fy * R[1 * 3 + j]);
    t[2 * 3 + j] = R[2 * 3 + j];
  }
  for (int i = 0; i < 3; ++i) {
    const double h0 = t[i * 3 + 0] * k->ifx, h1 = t[i * 3 + 1] * k->ify;
    imageBasis[i * 3 + 0] = (float)h0;
    imageBasis[i * 3 + 1] = (float)h1;
    imageBasis[i * 3 + 2] = (float)fmad(-h1, k->cy, fmad(-h0, k->cx, t[i * 3 + 2]));
  }
  const double ki[9] = {k->ifx, 0.0, -(k->cx * k->ifx), 0.0, k->ify, -(k->cy * k->ify), 0.0, 0.0, 1.0};
  for (int i = 0; i < 9; ++i) {
    kinv[i] = (float)ki[i];
    krlr[i] = (float)t[i];
  }
}

/* K Rt^-1 K^-1 and K t(Rt^-1) with the isometry inverse (RGBDOdometry.cpp:427-437) */
void orc_scalar_gn_params(const double* resultRt, const orc_kpre* k, float* krkinv, float* kt) {
  double Ri[9], ti[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ri[i * 3 + j] = resultRt[j * 4 + i];
  for (int i = 0; i < 3; ++i) ti[i] = -fmad(Ri[i * 3 + 2], resultRt[11], fmad(Ri[i * 3 + 1], resultRt[7], Ri[i * 3 + 0] * resultRt[3]));
  double M[9];
  for (int j = 0; j < 3; ++j) {
    M[0 * 3 + j] = fmad(k->cx, Ri[2 * 3 + j], k->fx * Ri[0 * 3 + j]);
    M[1 * 3 + j] = fmad(k->cy, Ri[2 * 3 + j], k->fy * Ri[1 * 3 + j]);
    M[2 * 3 + j] = Ri[2 * 3 + j];
  }
  for (int i = 0; i < 3; ++i) {
    const double h0 = M[i * 3 + 0] * k->ifx, h1 = M[i * 3 + 1] * k->ify;
    krkinv[i * 3 + 0] = (float)h0;
    krkinv[i * 3 + 1] = (float)h1;
    krkinv[i * 3 + 2] = (float)fmad(-h1, k->cy, fmad(-h0, k->cx, M[i * 3 + 2]));
  }
  kt[0] = (float)fmad(k->cx, ti[2], k->fx * ti[0]);
  kt[1] = (float)fmad(k->cy, ti[2], k->fy * ti[1]);
  kt[2] = (float)ti[2];
}

static void mul3f(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
static void mul3vf(const float* a, const float* v, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = a[i * 3 + 0] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

/* Eigen::Vector3f delta = jtj.ldlt().solve(jtr); R_lr = float(rodrigues(delta)) * R_lr (RGBDOdometry.cpp:371-377) */
void orc_scalar_so3_update(const float* jtj, const float* jtr, float* R_lr, double* resultR) {
  float delta[3];
  orc_scalar_ldlt_pivoted_f(3, jtj, jtr, delta, 1.0f / FLT_MAX);
  const double dd[3] = {(double)delta[0], (double)delta[1], (double)delta[2]};
  double rotUpdate[9];
  orc_scalar_rodrigues(dd, rotUpdate);
  float ru[9], nr[9];
  for (int i = 0; i < 9; ++i) ru[i] = (float)rotUpdate[i];
  mul3f(ru, R_lr, nr);
  for (int i = 0; i < 9; ++i) {
    R_lr[i] = nr[i];
    resultR[i] = (double)nr[i];
  }
}

/* RGBDOdometry.cpp:531-585 + OdometryProvider::computeUpdateSE3 */
void orc_scalar_gn_update(const float* A_icp, const float* b_icp, const float* A_rgb, const float* b_rgb, int icp, int rgb, float icpWeight,
                          const float* Rprev, const float* tprev, double* resultRt, double* A, double* b, float* Rcurr, float* tcurr) {
  const double w = (double)icpWeight, ww = w * w;
  for (int i = 0; i < 6; ++i) {
    for (int j = i; j < 6; ++j) {
      const double vi = icp ? (double)A_icp[i * 6 + j] : 0.0, vr = rgb ? (double)A_rgb[i * 6 + j] : 0.0;
      const double v = (icp && rgb) ? fmad(ww, vi, vr) : (icp ? vi : vr);
      A[j * 6 + i] = A[i * 6 + j] = v;
    }
    const double vi = icp ? (double)b_icp[i] : 0.0, vr = rgb ? (double)b_rgb[i] : 0.0;
    b[i] = (icp && rgb) ? fmad(w, vi, vr) : (icp ? vi : vr);
  }
  double x[6];
  if (!orc_scalar_ldlt_spd6(A, b, x)) orc_scalar_ldlt_pivoted_d(6, A, b, x, 1.0 / DBL_MAX);
  double R[9];
  orc_scalar_rodrigues(x + 3, R);
  double nr[16];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double v = fmad(R[i * 3 + 2], resultRt[2 * 4 + j], fmad(R[i * 3 + 1], resultRt[1 * 4 + j], R[i * 3 + 0] * resultRt[0 * 4 + j]));
      if (j == 3) v += x[i];
      nr[i * 4 + j] = v;
    }
  nr[12] = nr[13] = nr[14] = 0.0;
  nr[15] = 1.0;
  float Ro[9], to[3], RoT[9], ti[3], tc[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Ro[i * 3 + j] = (float)nr[i * 4 + j];
    to[i] = (float)nr[i * 4 + 3];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) RoT[i * 3 + j] = Ro[j * 3 + i];
  mul3vf(RoT, to, ti);
  ti[0] = -ti[0];
  ti[1] = -ti[1];
  ti[2] = -ti[2];
  mul3f(Rprev, RoT, Rcurr);
  mul3vf(Rprev, ti, tc);
  for (int i = 0; i < 3; ++i) tcurr[i] = tc[i] + tprev[i];
  memcpy(resultRt, nr, sizeof(nr));
}
